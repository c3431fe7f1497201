set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6a
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tl -- python3 $GRAFT_REPO_ROOT/tools/prof_rollout.py > $OUT/out.txt 2> $OUT/err.txt
cp $(ls $OUT/tl/*/*kernel_stats.csv | head -1) $OUT/stats.csv
head -14 $OUT/stats.csv | cut -c1-150
tail -3 $OUT/out.txt
