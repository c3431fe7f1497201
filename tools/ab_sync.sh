#!/bin/bash
# Same-box A/B of two builds on the per-rank iteration times (tools/sync_step.py, tools/sync_c5.py).  tools/ab_sync.sh <variant|default> <variant|default> [reps] [S list]
A=$1; B=$2; REPS=${3:-2}; SS=${4:-1,4,32}
libpath() { if [ "$1" = default ]; then echo ""; else echo "$PWD/ffvd_amd/libffvd_hip_$1.so"; fi; }
for i in $(seq $REPS); do
  for v in $A $B; do
    L=$(libpath $v)
    if [ -n "$L" ]; then export FFVD_LIB=$L; else unset FFVD_LIB; fi
    python3 tools/sync_step.py S=$SS 2>/dev/null | sed "s/^/$v rep $i /"
    python3 tools/sync_c5.py 2>/dev/null | head -1 | sed "s/^/$v rep $i /"
  done
done
