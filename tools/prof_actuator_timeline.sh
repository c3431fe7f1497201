set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/actuator_tl_${1:-both}
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -- python3 $GRAFT_REPO_ROOT/tools/actuator_step.py $1 > $OUT/out.txt 2> $OUT/err.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "prep_hypers" in r["Kernel_Name"]]
i0, i1 = starts[-2], starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
busy = 0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    print("%8.1f %7.1f us q%-2s grid %7s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Grid_Size_X"], r["Kernel_Name"][:60]))
print("kernels", i1 - i0, "sum of durations %.1f us, span %.1f us" % (busy / 1e3, (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
PY
