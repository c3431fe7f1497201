#!/bin/bash
# Same-box A/B of the product library against a variant build on the per-rank iteration times.  tools/ab_lib.sh <variant> [reps] [S list]
V=$1; REPS=${2:-2}; SS=${3:-1,4,32}
for i in $(seq $REPS); do
  for v in default $V; do
    if [ "$v" = default ]; then unset FFVD_LIB; else export FFVD_LIB=$PWD/ffvd_amd/libffvd_hip_$v.so; fi
    python3 tools/sync_step.py S=$SS 2>/dev/null | sed "s/^/$v rep $i /"
    python3 tools/sync_c5.py 2>/dev/null | head -1 | sed "s/^/$v rep $i /"
  done
done
