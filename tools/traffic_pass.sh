#!/bin/bash
# On the GPU box: the two HBM-traffic counter passes (FETCH_SIZE, WRITE_SIZE) of one bench workload, summarised.
# usage: tools/traffic_pass.sh <outdir> <tag> <workload> [bench args...]  -> <outdir>/<tag>_pmc_traffic.md, <outdir>/traffic_<workload>.json
set -u
OUT=$1; TAG=$2; WL=$3; shift 3
mkdir -p "$OUT"
PMC_ONLY="5 6" tools/pmc_passes.sh "$OUT/${TAG}_pmc" --workload $WL "$@" --set fusePairs=0 --no-cpu-baseline --no-kernel-timing || exit 1
python3 tools/pmc_summary.py "$OUT/${TAG}_pmc" > "$OUT/${TAG}_pmc_traffic.md"
python3 tools/traffic_json.py "$OUT/${TAG}_pmc_traffic.md" $WL "$OUT/traffic_$WL.json"
rm -rf "$OUT/${TAG}_pmc"/pass*/
