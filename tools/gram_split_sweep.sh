#!/bin/bash
# Per-rank iteration time at config 2's shape for S chains (= a rank's share on 32 / S GPUs) against the number of row ranges of the
# Gram launch (FFVD_GSPLIT; "auto" = gram_ksplit's own choice).   tools/gram_split_sweep.sh "1,2,4,8,16" "auto 1 2 3 4 8"
SS=${1:-1,2,4,8,12,16,20,24}
GS=${2:-auto 1 2 3 4 6 8}
for g in $GS; do
  if [ "$g" = auto ]; then unset FFVD_GSPLIT; else export FFVD_GSPLIT=$g; fi
  python3 tools/sync_step.py S=$SS 2>/dev/null | sed "s/^/gsplit=$g /"
done
