#!/bin/bash
# Same-box A/B of the dataflow Cholesky's small-batch modes: FFVD_DF_FINE = 1 (per-column progress words + early S_rr sums, the default
# when every block row has a CU of its own), 2 (progress words only), 0 (row-level progress).  tools/ab_df_fine.sh [reps] [S list]
REPS=${1:-2}; SS=${2:-1,2,4}
for i in $(seq $REPS); do
  for v in 1 2 0; do
    export FFVD_DF_FINE=$v
    python3 tools/sync_step.py S=$SS 2>/dev/null | sed "s/^/fine=$v rep $i /"
    python3 tools/sync_c5.py 2>/dev/null | head -1 | sed "s/^/fine=$v rep $i /"
  done
done
