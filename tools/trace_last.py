"""Print the kernel timeline of the last N kernels from a rocprofv3 kernel_trace.csv (tools helper)."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 70
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-40s q%-3s start %8.1f dur %7.1f  grid %s" % (r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id", "?")[-3:], (s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size", "")))
