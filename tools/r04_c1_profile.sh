#!/bin/bash
# Round-4 artefacts of BASELINE configs[0] (the one-launch iteration) in one GPU call: per-call latencies, the bench line, kernel stats,
# phase stamps of the debug build.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4c1
mkdir -p $OUT
cd $R
( python3 tools/actuator_step.py 2>/dev/null; ACT_S=10 python3 tools/actuator_step.py 2>/dev/null; FFVD_NO_TINY=1 python3 tools/actuator_step.py 2>/dev/null | sed 's/^/FFVD_NO_TINY=1 /'; FFVD_NO_TINY=1 ACT_S=10 python3 tools/actuator_step.py 2>/dev/null | sed 's/^/FFVD_NO_TINY=1 /' ) > $OUT/actuator_step.txt
cat $OUT/actuator_step.txt
python3 bench.py --workload c1 > $OUT/bench_c1.json 2>/dev/null
( for m in forward train; do for s in 1 10; do python3 tools/tiny_trace.py $m $s 2>/dev/null; echo; done; done ) > $OUT/tiny_trace.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --workload c1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/prof.err || true
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -4 $OUT/kernel_stats.csv
