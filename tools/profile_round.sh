#!/bin/bash
# On the GPU box: headline bench + kernel trace + HBM-traffic counter pass of the same command.
# usage: tools/profile_round.sh <outdir> <tag> [bench args...]   (writes <outdir>/<tag>_bench.json, _kernel_stats.csv, _pmc_*.md)
set -u
OUT=$1; TAG=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd "$ROOT"
timeout -k 10 400 python bench.py "$@" > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err" || { echo "bench failed"; tail -5 "$OUT/${TAG}_bench.err"; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_trace" -o t -- python3 "$ROOT/bench.py" "$@" --no-cpu-baseline > "$OUT/${TAG}_trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/${TAG}_trace.log"; exit 1; }
python3 "$ROOT/tools/kstats.py" "$OUT/${TAG}_trace/t_results.db" "$OUT/${TAG}_kernel_stats_overlapped.csv" > "$OUT/${TAG}_kernel_stats_overlapped.txt"
# the same command with one pass stream and unfused launches: every kernel by itself (the durations roofline.achieved is computed from)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_trace1" -o t -- python3 "$ROOT/bench.py" "$@" --set passStreams=1 --set fusePairs=0 --no-cpu-baseline > "$OUT/${TAG}_trace1.log" 2>&1 || { echo "trace1 failed"; tail -5 "$OUT/${TAG}_trace1.log"; exit 1; }
python3 "$ROOT/tools/kstats.py" "$OUT/${TAG}_trace1/t_results.db" "$OUT/${TAG}_kernel_stats.csv" > "$OUT/${TAG}_kernel_stats.txt"
rm -rf "$OUT/${TAG}_trace1"
cd "$ROOT"
PMC_ONLY="5 6" tools/pmc_passes.sh "$OUT/${TAG}_pmc" "$@" --set fusePairs=0 --no-cpu-baseline --no-kernel-timing || exit 1
python3 tools/pmc_summary.py "$OUT/${TAG}_pmc" > "$OUT/${TAG}_pmc_traffic.md"
rm -rf "$OUT/${TAG}_pmc"/pass*/ "$OUT/${TAG}_trace"
cat "$OUT/${TAG}_kernel_stats.txt"
