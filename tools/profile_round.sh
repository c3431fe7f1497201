#!/bin/bash
# Round profile (run on the GPU box through gpurun): kernel-trace stats and PMC passes of bench.py for one workload.
#   tools/profile_round.sh <tag> <commit> <key> <gram_kernel_prefix> <proj_kernel_prefix> -- <bench.py args...>
# Outputs under gpurun_out/<tag>/ ; copy the summaries into profiles/.
set -e
TAG=$1; COMMIT=$2; KEY=$3; GP=$4; PP=$5; shift 6
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py "$@" --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py "$@" --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
SHORT="$@ --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $SHORT > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $SHORT > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $SHORT > /dev/null 2> $OUT/pmc_sq.err
python3 $R/tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_summary.txt $OUT/traffic.json "$KEY" "$COMMIT" "$GP" "$PP" > /dev/null
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
head -c 600 $OUT/bench.json; echo; grep -E "gram|proj_gemm|kfu|potrf" $OUT/pmc_summary.txt | cut -c1-200
