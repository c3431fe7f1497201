#!/bin/bash
# kernel durations of a short particle-Gibbs sweep (GPU box): tools/pg_trace.sh <tag> ; FFVD_PG_FUSED selects the form
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pg -- python3 $GRAFT_REPO_ROOT/tools/pg_time.py > $OUT/pg.out 2> $OUT/pg.err
cp $(ls $OUT/pg/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv | cut -c1-150
