#!/usr/bin/env python3
"""Cache-level figures per kernel from a tools/pmc_summary.py table (passes 2, 3, 4 of tools/pmc_passes.sh): share of the
L1 (TCP) accesses that go on to L2 (TCC), L2 miss rate, L2 fill requests (TCC_EA0_RDREQ: to the Infinity Cache / HBM) per
dispatch, VALU instructions per vector-memory read instruction, VALU lane utilisation (thread-cycles / (64 x active cycles)).
usage: tools/pmc_cache.py <summary.md>  (prints a markdown table)"""
import sys
rows = [l.strip().strip("|").split("|") for l in open(sys.argv[1]) if l.startswith("|")]
hdr = [h.strip() for h in rows[0]]
print("| kernel | dispatches | TCP accesses | TCP -> TCC read requests (share of accesses) | TCC hit | TCC miss (rate) | TCC_EA0_RDREQ per dispatch | VALU insts | SALU insts | VMEM read insts | VALU / VMEM | VALU lane utilisation |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows[2:]:
    d = dict(zip(hdr, [x.strip() for x in r]))

    def f(k):
        try:
            return float(d.get(k, "nan"))
        except ValueError:
            return float("nan")
    if not f("TCP_TOTAL_CACHE_ACCESSES_sum") > 0 or "rocclr" in d["kernel"]:
        continue
    n = max(1, int(d["dispatches"]))
    print("| %s | %d | %.3g | %.3g (%.3f) | %.3g | %.3g (%.3f) | %.3g | %.3g | %.3g | %.3g | %.1f | %.3f |" % (
        d["kernel"].replace("rt::", ""), n, f("TCP_TOTAL_CACHE_ACCESSES_sum"), f("TCP_TCC_READ_REQ_sum"),
        f("TCP_TCC_READ_REQ_sum") / f("TCP_TOTAL_CACHE_ACCESSES_sum"), f("TCC_HIT_sum"), f("TCC_MISS_sum"),
        f("TCC_MISS_sum") / max(1.0, f("TCC_HIT_sum") + f("TCC_MISS_sum")), f("TCC_EA0_RDREQ_sum") / n,
        f("SQ_INSTS_VALU"), f("SQ_INSTS_SALU"), f("SQ_INSTS_VMEM_RD"), f("SQ_INSTS_VALU") / max(1.0, f("SQ_INSTS_VMEM_RD")),
        f("SQ_THREAD_CYCLES_VALU") / max(1.0, 64 * f("SQ_ACTIVE_INST_VALU"))))
