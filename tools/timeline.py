"""Print the kernel timeline of the last iteration from a rocprofv3 rocpd database (tools helper)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 70
last = rows[-n:]
t0 = last[0][1]
for name, s, e in last:
    print("%-44s start %9.1f dur %8.1f" % (name.split("(")[0][-44:], (s - t0) / 1e3, (e - s) / 1e3))
