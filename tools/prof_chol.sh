#!/bin/bash
# per-launch durations of the Cholesky kernels, left-looking vs right-looking, config 2 (run on the GPU box)
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ll -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/ll.json 2> $OUT/ll.err
FFVD_CHOL_RIGHT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/rl.json 2> $OUT/rl.err
python3 - <<PY
import csv, glob, collections
for v in ("ll", "rl"):
    f = glob.glob("$OUT/%s/**/*kernel_trace.csv" % v, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "potrf" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # last iteration's launches with a big grid (the 128-matrix factorisation)
    big = [r for r in rows if int(r["Grid_Size_X"]) >= 128 * 256]
    print(v, "launches", len(rows), "big", len(big))
    for r in big[-20:]:
        print("  %-40s grid %7s  %8.1f us" % (r["Kernel_Name"][:40], r["Grid_Size_X"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
