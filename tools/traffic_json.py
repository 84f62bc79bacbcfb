#!/usr/bin/env python3
"""HBM traffic per launch of the closest-hit kernels from a rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run
(tools/pmc_passes.sh passes 5 and 6, summarised by tools/pmc_summary.py).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests of 16-B-per-lane loads at 64 B, so it
is doubled (MI355X_MICROARCH.md, HBM section) for the traversal kernels, whose traffic is 16-byte-per-lane record loads;
WRITE_SIZE is exact for 16-B-per-lane stores.  Other access widths are uncalibrated (same section): k_shade (8 / 12 / 16-byte
accesses mixed) and k_resolve (4- and 12-byte loads since round 3) are reported with the raw counter, marked as such — doubled,
k_resolve's figure exceeds what HBM can deliver in the kernel's duration.
usage: tools/traffic_json.py <pmc_traffic.md> <workload> <out.json>"""
import json
import sys

rows = [l.strip().strip("|").split("|") for l in open(sys.argv[1]) if l.startswith("|")]
hdr = [h.strip() for h in rows[0]]
out = {"workload": sys.argv[2], "source": sys.argv[1], "unit": "bytes per launch",
       "fetch_correction": "2.0 for k_extend / k_shadow (16-B-per-lane loads), 1.0 elsewhere (uncalibrated widths)", "kernels": {}}
agg = {}
for r in rows[2:]:
    r = [x.strip() for x in r]
    d = dict(zip(hdr, r))
    name = d["kernel"]
    cls = "k_extend" if "k_extend" in name else "k_shadow" if "k_shadow" in name else name.replace("rt::", "")
    fetch, write = float(d["FETCH_SIZE"]), float(d["WRITE_SIZE"])
    if fetch != fetch or write != write:      # a kernel that only one of the two counter passes launched (the per-scene kernel
        continue                              # trial can settle differently from run to run): no pair of figures, skipped
    a = agg.setdefault(cls, {"launches": 0, "fetch_kib": 0.0, "write_kib": 0.0, "fetch_correction": 2.0 if cls in ("k_extend", "k_shadow") else 1.0})
    a["launches"] += int(d["dispatches"])
    a["fetch_kib"] += fetch
    a["write_kib"] += write
for cls, a in agg.items():
    if a["launches"]:
        a["hbm_bytes_per_launch"] = round((a["fetch_correction"] * a["fetch_kib"] + a["write_kib"]) * 1024.0 / a["launches"], 1)
        out["kernels"][cls] = a
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"].get("k_extend")))
