#!/bin/bash
# Round-4 artefacts in one GPU call: c2 profile with PMC passes (traffic.json at HEAD), bench lines, next rows, per-rank step times at
# config 2 (chains) and config 5 (latent dims).
set -e
R=$GRAFT_REPO_ROOT
COMMIT=$1
cd $R
tools/profile_round.sh r4_c2 $COMMIT c2/f64/gram gram_kernel kfu_build --
OUT=$R/gpurun_out/r4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python3 $R/bench.py --workload c5 --no-cpu-baseline > $OUT/bench_c5.json 2>/dev/null
python3 $R/bench.py --route reference --no-cpu-baseline > $OUT/bench_c2_reference.json 2>/dev/null
python3 $R/tools/bench_next.py > $OUT/next_rows.json 2>/dev/null
python3 $R/tools/sync_step.py S=1,2,4,8,16,32 > $OUT/per_rank_step.txt 2>/dev/null
python3 $R/tools/sync_c5.py >> $OUT/per_rank_step.txt 2>/dev/null
cat $OUT/per_rank_step.txt
head -c 400 $OUT/bench_c2.json; echo
