#!/bin/bash
# GPU box: the round's profiles for the three traced workloads + SQ/TCP counters of the final kernels (C2, 128 spp, one stream).
# usage: tools/profile_all.sh <outdir> <tag>     e.g. tools/profile_all.sh gpurun_out/r2 r2
set -u
OUT=$1; TAG=$2
mkdir -p "$OUT"
tools/profile_round.sh "$OUT" ${TAG}_headline --steps 2 --warmup 2 > "$OUT/${TAG}_headline.log" 2>&1 || { tail -5 "$OUT/${TAG}_headline.log"; exit 1; }
python3 tools/traffic_json.py "$OUT/${TAG}_headline_pmc_traffic.md" c2 "$OUT/traffic_c2.json"
tools/profile_round.sh "$OUT" ${TAG}_c3 --workload c3 --steps 3 --warmup 2 > "$OUT/${TAG}_c3.log" 2>&1 || { tail -5 "$OUT/${TAG}_c3.log"; exit 1; }
python3 tools/traffic_json.py "$OUT/${TAG}_c3_pmc_traffic.md" c3 "$OUT/traffic_c3.json"
tools/profile_round.sh "$OUT" ${TAG}_c4 --workload c4 --steps 2 --warmup 1 > "$OUT/${TAG}_c4.log" 2>&1 || { tail -5 "$OUT/${TAG}_c4.log"; exit 1; }
python3 tools/traffic_json.py "$OUT/${TAG}_c4_pmc_traffic.md" c4 "$OUT/traffic_c4.json"
# SQ / TCP / TA counters of the kernels that run today, one stream, unfused launches, 128 spp of the headline frame
PMC_ONLY="1 2 4 7" tools/pmc_passes.sh "$OUT/${TAG}_sq" --spp 128 --steps 1 --warmup 1 --prepare-frames 5 --set passStreams=1 --set fusePairs=0 --set kernelChoice=1 --no-cpu-baseline --no-kernel-timing > "$OUT/${TAG}_sq.log" 2>&1 || { tail -5 "$OUT/${TAG}_sq.log"; exit 1; }
python3 tools/pmc_summary.py "$OUT/${TAG}_sq" > "$OUT/${TAG}_pmc_c2_128spp_single_stream.md"
{ echo; echo "Derived (tools/pmc_derive.py; kernel set pinned with kernelChoice=1 so that every counter pass runs the same launches):"; echo; python3 tools/pmc_derive.py "$OUT/${TAG}_pmc_c2_128spp_single_stream.md"; } >> "$OUT/${TAG}_pmc_c2_128spp_single_stream.md"
rm -rf "$OUT/${TAG}_sq"/pass*/
for w in c1 c5; do timeout -k 10 400 python bench.py --workload $w --steps 2 --warmup 1 > "$OUT/${TAG}_${w}_bench.json" 2> "$OUT/${TAG}_${w}_bench.err" || { echo "$w bench failed"; tail -3 "$OUT/${TAG}_${w}_bench.err"; }; done
ls "$OUT"
