set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG:-r3final}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python3 $R/bench.py --workload c4 > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python3 $R/bench.py --route reference --no-cpu-baseline > $OUT/bench_c2_reference.json 2>/dev/null
python3 $R/bench.py --workload c5 --no-cpu-baseline > $OUT/bench_c5.json 2>/dev/null
python3 $R/tools/bench_next.py > $OUT/next_rows.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -- python3 $R/tools/prof_grad.py > /dev/null 2> $OUT/train.err
cp $(ls $OUT/train/*/*kernel_stats.csv | head -1) $OUT/train_step_kernel_stats.csv
rm -rf $OUT/train
python3 $R/tools/sync_step.py S=1,2,4,8,16,32 > $OUT/sync_step.txt 2>/dev/null
cat $OUT/sync_step.txt
head -c 300 $OUT/bench_c2.json; echo
