#!/bin/bash
# Per-kernel times of a training step with and without an environment switch (run on the GPU box):
#   tools/grad_env_ab.sh <tag> <VAR=VALUE> [reps]
set -e
R=$GRAFT_REPO_ROOT
TAG=$1; KV=$2; REPS=${3:-2}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
for i in $(seq $REPS); do
for mode in base switched; do
  D=$R/gpurun_out/$TAG/${mode}_$i
  if [ $mode = switched ]; then export "$KV"; else unset "${KV%%=*}"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/prof_grad.py > $D.out 2> $D.err
  python3 - <<PY
import csv, glob
f = glob.glob("$D/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("== $mode $i ($KV)")
for r in rows[:7]:
    print("  %-60s calls %4s avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
done
