#!/bin/bash
# Gram tile pass + combine pass durations at one chain of config 2 against the number of row ranges (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
for g in ${1:-3 4 6 8 12}; do
  export FFVD_GSPLIT=$g
  rm -rf /tmp/gs_$g
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gs_$g -- python3 $GRAFT_REPO_ROOT/tools/sync_step.py S=${SYNC_S:-1} > /tmp/gs_$g.out 2>/dev/null
  f=$(find /tmp/gs_$g -name "*kernel_stats.csv" | head -1)
  echo "gsplit=$g $(grep SYNC /tmp/gs_$g.out)"
  grep -E "gram_kernel<1>|gram_combine|potrf_df" $f | awk -F, '{printf "   %-60s calls %s avg %.1f us\n", substr($1,1,60), $2, $4/1000}'
done
