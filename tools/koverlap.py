#!/usr/bin/env python3
"""Concurrency census of a rocprofv3 rocpd kernel trace: tools/koverlap.py <results.db> [t0_fraction t1_fraction]
Prints, for the chosen window of the trace, how long 0, 1, 2, ... kernels were running at the same time, the per-kernel
mean duration, and per-stream/queue launch counts."""
import re, sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute("select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")).fetchall()
T0, T1 = rows[0][1], max(r[2] for r in rows)
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
w0, w1 = T0 + (T1 - T0) * f0, T0 + (T1 - T0) * f1
ev = []
per = defaultdict(lambda: [0, 0.0])
qs = defaultdict(int)
for r in rows:
    s, e = r[1], r[2]
    if e <= w0 or s >= w1:
        continue
    ev.append((max(s, w0), 1)); ev.append((min(e, w1), -1))
    short = re.sub(r"\(.*", "", r[0]).replace("void rt::", "").replace("rt::", "")[:40]
    per[short][0] += 1; per[short][1] += (e - s) / 1e3
    if qcol:
        qs[r[3]] += 1
ev.sort()
busy = defaultdict(float)
cur, last = 0, w0
for t, d in ev:
    busy[cur] += t - last
    last = t
    cur += d
busy[cur] += w1 - last
tot = w1 - w0
print("window %.2f ms; columns used: %s" % (tot / 1e6, qcol))
for k in sorted(busy):
    print("  %d kernels running: %8.2f ms (%5.1f %%)" % (k, busy[k] / 1e6, 100 * busy[k] / tot))
for k, (n, us) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("  %-40s %5d launches, mean %9.1f us, total %9.2f ms" % (k, n, us / n, us / 1e3))
if qcol:
    print("  launches per %s: %s" % (qcol, dict(qs)))
