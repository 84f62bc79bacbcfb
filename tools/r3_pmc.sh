#!/bin/bash
# On the GPU box: SQ / TCC / TCP / TA counter passes of one workload with the kernel set pinned, every kernel by itself
# (one pass stream, no fused launches).  usage: tools/r3_pmc.sh <outdir> <tag> <workload> <kernelChoice> [extra bench args]
set -u
OUT=$1; TAG=$2; WL=$3; KC=$4; shift 4
PMC_ONLY="${PMC_ONLY:-1 2 3 4 7}" tools/pmc_passes.sh "$OUT/${TAG}_pmc" --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing \
    --set kernelChoice=$KC --set passStreams=1 --set fusePairs=0 "$@" || exit 1
python3 tools/pmc_summary.py "$OUT/${TAG}_pmc" > "$OUT/${TAG}_pmc.md"
python3 tools/pmc_derive.py "$OUT/${TAG}_pmc.md" > "$OUT/${TAG}_pmc_derived.md"
rm -rf "$OUT/${TAG}_pmc"/pass*/
cat "$OUT/${TAG}_pmc_derived.md"
