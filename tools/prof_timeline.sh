#!/bin/bash
# kernel timeline of the last iteration (start offset, duration, name) (run on the GPU box):
#   tools/prof_timeline.sh <tag> [bench.py args]          config 2 through bench.py
#   SYNC_S=4 tools/prof_timeline.sh <tag>                 one rank's share (S chains) through tools/sync_step.py
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
shift
mkdir -p $OUT
if [ -n "$SYNC_S" ]; then
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -- python3 $GRAFT_REPO_ROOT/tools/sync_step.py S=$SYNC_S > $OUT/tl.json 2> $OUT/tl.err
else
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/tl.json 2> $OUT/tl.err
fi
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last iteration = from the last prep_hypers launch that is followed by a finalize
starts = [i for i, r in enumerate(rows) if "prep_hypers" in r["Kernel_Name"]]
i0 = starts[-1]
fin = [i for i, r in enumerate(rows) if "finalize_kernel" in r["Kernel_Name"] and i > i0]
if not fin: i0 = starts[-2]; fin = [i for i, r in enumerate(rows) if "finalize_kernel" in r["Kernel_Name"] and i > i0]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:fin[0] + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f  %8.1f us  q%-3s grid %8s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Grid_Size_X"], r["Kernel_Name"][:70]))
PY
