#!/usr/bin/env python3
"""VALU / texture-addresser busy fractions and resident waves per SIMD of every kernel, from a tools/pmc_summary.py table
(passes 1 and 7 of tools/pmc_passes.sh at least), as the JSON bench.py reads for roofline.valu_busy / ta_busy.
Same arithmetic as tools/pmc_derive.py: duration of a kernel's dispatches = GRBM_TA_BUSY / 8 XCDs (clocks); SQ_* cycle
counters are quad-cycles summed over waves / SIMDs; 1024 SIMDs.
usage: tools/pmc_units.py <summary.md> <workload> <out.json> [source note]"""
import json
import sys

rows = [l.strip().strip("|").split("|") for l in open(sys.argv[1]) if l.startswith("|")]
hdr = [h.strip() for h in rows[0]]
out = {"workload": sys.argv[2], "source": sys.argv[4] if len(sys.argv) > 4 else sys.argv[1], "kernels": {}}
for r in rows[2:]:
    d = dict(zip(hdr, [x.strip() for x in r]))

    def f(k):
        try:
            return float(d.get(k, "nan"))
        except ValueError:
            return float("nan")
    dur = f("GRBM_TA_BUSY")
    if not dur == dur or dur <= 0:
        continue
    name = d["kernel"].replace("rt::", "")
    short = name.split("<")[0]
    rec = {"dispatches": int(d["dispatches"]), "valu_busy": round(f("SQ_ACTIVE_INST_VALU") / (32 * dur), 3),
           "waves_per_simd": round(f("SQ_WAVE_CYCLES") / (32 * dur), 2),
           "wait_share": round(f("SQ_WAIT_ANY") / max(1.0, f("SQ_WAVE_CYCLES")), 3),
           "ta_busy": round(f("TA_BUSY_avr") / (dur / 8), 3) if f("TA_BUSY_avr") == f("TA_BUSY_avr") else None}
    out["kernels"][name] = rec
    # the family entry bench.py looks up: the variant with the most dispatches (the timed one, not the instrumented one)
    if short not in out["kernels"] or out["kernels"][short]["dispatches"] < rec["dispatches"]:
        out["kernels"][short] = dict(rec, variant=name)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in out["kernels"].items() if "<" not in k}))
