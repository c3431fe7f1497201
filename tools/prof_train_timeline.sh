#!/bin/bash
# kernel timeline of the last training step (forward + backward + Adam) at config 2 (run on the GPU box)
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -- python3 $GRAFT_REPO_ROOT/tools/prof_grad.py > $OUT/tl.json 2> $OUT/tl.err
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "prep_hypers" in r["Kernel_Name"]]
i0 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f  %8.1f us  q%-3s grid %8s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Grid_Size_X"], r["Kernel_Name"][:64]))
PY
