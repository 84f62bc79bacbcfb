#!/usr/bin/env python3
"""Launch-by-launch durations of the last N kernel launches of a rocprofv3 rocpd database (kernel-trace), in start order:
tools/ktimeline.py <results.db> [N]   (one pass of a frame rendered with passStreams=1 = 3 * maxBounces + 2 launches)"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = db.execute("select name, start, end from kernels order by start").fetchall()
rows = rows[-n:]
t0 = rows[0][1]
for name, s, e in rows:
    short = re.sub(r"\(.*", "", name).replace("void rt::", "").replace("rt::", "")
    print("%9.1f us  +%8.1f us  gap-free dur %8.1f us  %s" % ((s - t0) / 1e3, 0.0, (e - s) / 1e3, short[:60]))
