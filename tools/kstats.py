#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 rocpd database (kernel-trace): tools/kstats.py <results.db> [csv_out]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
for r in rows:
    lines.append('"%s",%d,%d,%.1f,%.2f,%d,%d' % (r[0], r[1], r[2], r[3], 100.0 * r[2] / tot, r[4], r[5]))
    print("%-70s %5d %10.2f ms  avg %9.1f us  min %8.1f  max %9.1f  %5.1f%%" % (r[0][:70], r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot))
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("\n".join(lines) + "\n")
