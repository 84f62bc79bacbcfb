#!/usr/bin/env python3
"""Summarises the counter_collection.csv files written by tools/pmc_passes.sh: per kernel (short name), sum of each
counter over all dispatches.  Usage: tools/pmc_summary.py <outdir> > profiles/<name>.md"""
import csv
import glob
import re
import sys
from collections import defaultdict

out = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
percall = defaultdict(int)
seen = set()
for f in sorted(glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            percall[(f, k)] += 1
for (f, k), n in percall.items():
    calls[k] = max(calls[k], n)       # every pass runs the same command: dispatches per kernel = count in one pass
names = sorted({c for k in tot for c in tot[k]})
print("| kernel | dispatches | " + " | ".join(names) + " |")
print("|---|---|" + "---|" * len(names))
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0)):
    print("| %s | %d | " % (k, calls[k]) + " | ".join("%.4g" % tot[k].get(c, float("nan")) for c in names) + " |")
