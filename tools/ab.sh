#!/bin/bash
# Same-box A/B of two builds of the library: alternating bench runs, `reps` each, one line per run with the iteration time and the
# stage times.   tools/ab.sh <variant-of-ffvd_amd/build.py:VARIANTS | default> <variant | default> [reps] [bench args...]
# e.g.  tools/ab.sh default diag64 3 --steps 40
A=$1; B=$2; REPS=${3:-3}; shift 3
OUT=${AB_OUT:-gpurun_out/ab.txt}
mkdir -p "$(dirname "$OUT")"
libpath() { if [ "$1" = default ]; then echo ""; else echo "$PWD/ffvd_amd/libffvd_hip_$1.so"; fi; }
for v in $A $B; do
  if [ "$v" != default ] && [ ! -f "$(libpath $v)" ]; then python -m ffvd_amd.build --$v > /dev/null || exit 1; fi
done
for i in $(seq $REPS); do
  for v in $A $B; do
    L=$(libpath $v)
    if [ -n "$L" ]; then export FFVD_LIB=$L; else unset FFVD_LIB; fi
    python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
st = d['roofline']['stage_ms_per_step']
print('$v', 'rep', $i, 'ms_per_step %.4f' % d['ms_per_step'], 'median %.4f' % d['median_ms_per_step'], 'min %.4f' % d['min_ms_per_step'],
      ' '.join('%s %.4f' % (k, v) for k, v in st.items()), 'nll %.15g' % d['nll'], flush=True)" | tee -a "$OUT" || exit 1
  done
done
