#!/usr/bin/env python3
"""One frame of a rocprofv3 rocpd kernel trace, launch by launch and queue by queue: tools/kframe.py <results.db> [frame_index]
A frame runs from a k_init_frame launch to the next k_finalize; frame_index counts from the start (default: the third-last,
a timed frame of bench.py).  Prints the concurrency census of the frame and, per queue, start / duration of every launch."""
import re, sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = db.execute("select name, start, end, %s from kernels order by start" % qcol).fetchall()
inits = [r[1] for r in rows if "k_init_frame" in r[0]]
fins = [r[2] for r in rows if "k_finalize" in r[0]]
frames = []
for t in inits:
    f = [x for x in fins if x > t]
    if f: frames.append((t, f[0]))
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(frames) - 3
t0, t1 = frames[k]
print("frame %d of %d: %.2f ms" % (k, len(frames), (t1 - t0) / 1e6))
sel = [r for r in rows if r[1] >= t0 and r[2] <= t1]
ev = []
for r in sel: ev.append((r[1], 1)); ev.append((r[2], -1))
ev.sort()
busy = defaultdict(float); cur = 0; last = t0
for t, d in ev:
    busy[cur] += t - last; last = t; cur += d
for c in sorted(busy): print("  %d kernels running: %7.2f ms" % (c, busy[c] / 1e6))
per = defaultdict(lambda: [0, 0.0])
for r in sel:
    short = re.sub(r"\(.*", "", r[0]).replace("void rt::", "").replace("rt::", "")[:44]
    per[short][0] += 1; per[short][1] += (r[2] - r[1]) / 1e6
for s, (n, ms) in sorted(per.items(), key=lambda kv: -kv[1][1]): print("  %-44s %4d launches %8.2f ms" % (s, n, ms))
if len(sys.argv) > 3:
    for q in sorted(set(r[3] for r in sel)):
        print("queue", q)
        for r in sel:
            if r[3] == q:
                short = re.sub(r"\(.*", "", r[0]).replace("void rt::", "").replace("rt::", "")[:40]
                print("  %9.3f ms  %8.3f ms  %s" % ((r[1] - t0) / 1e6, (r[2] - r[1]) / 1e6, short))
