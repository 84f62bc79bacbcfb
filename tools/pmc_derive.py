#!/usr/bin/env python3
"""Derived figures from a tools/pmc_summary.py table (passes 1, 2, 4, 7 of tools/pmc_passes.sh): per kernel VALU busy,
mean resident waves per SIMD, wave-time split, LDS bank-conflict rate, vector-memory figures.
Duration of a kernel's dispatches = GRBM_TA_BUSY / 8 XCDs (clocks); SQ_* cycle counters are in quad-cycles summed over
waves / SIMDs (MI355X_MICROARCH.md, per-instruction constants); 1024 SIMDs.
usage: tools/pmc_derive.py <summary.md>  (prints a markdown table)"""
import sys
rows = [l.strip().strip("|").split("|") for l in open(sys.argv[1]) if l.startswith("|")]
hdr = [h.strip() for h in rows[0]]
print("| kernel | dispatches | VALU busy | waves / SIMD | waiting (s_waitcnt) | issue-stalled | issuing | LDS bank-conflict cycles / LDS active | VMEM read insts / wave | TA busy | TCP pending-stall / TA busy |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows[2:]:
    d = dict(zip(hdr, [x.strip() for x in r]))
    f = lambda k: float(d.get(k, "nan"))
    dur = f("GRBM_TA_BUSY")
    if not dur == dur or dur <= 0:
        continue
    wc = f("SQ_WAVE_CYCLES")
    print("| %s | %s | %.2f | %.2f | %.2f | %.2f | %.2f | %.3f | %.0f | %.2f | %.2f |" % (
        d["kernel"].replace("rt::", ""), d["dispatches"], f("SQ_ACTIVE_INST_VALU") / (32 * dur), wc / (32 * dur),
        f("SQ_WAIT_ANY") / wc, f("SQ_WAIT_INST_ANY") / wc, f("SQ_ACTIVE_INST_ANY") / wc,
        f("SQ_LDS_BANK_CONFLICT") / max(1.0, f("SQ_LDS_IDX_ACTIVE")), f("SQ_INSTS_VMEM_RD") / max(1.0, f("SQ_WAVES")),
        f("TA_BUSY_avr") / (dur / 8) if f("TA_BUSY_avr") == f("TA_BUSY_avr") else float("nan"),
        f("TCP_PENDING_STALL_CYCLES_sum") / max(1.0, f("TA_TA_BUSY_sum"))))
