#!/bin/bash
# Per-kernel times of a training step for several builds of the library (run on the GPU box):
#   tools/grad_kernel_ab.sh <tag> <variant|default> [<variant> ...]      (variants = ffvd_amd/libffvd_hip_<variant>.so)
set -e
R=$GRAFT_REPO_ROOT
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
for v in "$@"; do
  if [ "$v" = default ]; then unset FFVD_LIB; else export FFVD_LIB=$R/ffvd_amd/libffvd_hip_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/$v -- python3 $R/tools/prof_grad.py > $R/gpurun_out/$TAG/$v.out 2> $R/gpurun_out/$TAG/$v.err
  python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/$TAG/$v/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("== $v")
for r in rows[:8]:
    print("  %-60s calls %4s avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
