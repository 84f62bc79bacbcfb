#!/bin/bash
# On the GPU box: launch-by-launch durations of one pass (one stream, unfused launches) + the queue sizes per bounce.
# usage: tools/bounce_timeline.sh <outdir> <workload> <kernelChoice> [bench args...]
set -u
OUT=$1; WL=$2; KC=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
RT_DEBUG=1 timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT/tl_$WL" -o t -- python3 "$ROOT/bench.py" --workload $WL --steps 1 --warmup 0 --prepare-frames 1 \
  --set passStreams=1 --set fusePairs=0 --set kernelChoice=$KC --no-cpu-baseline "$@" > "$OUT/tl_$WL.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/tl_$WL.log"; exit 1; }
python3 "$ROOT/tools/ktimeline.py" "$OUT/tl_$WL/t_results.db" 30 > "$OUT/timeline_$WL.txt"
grep "queue sizes" "$OUT/tl_$WL.log" | tail -1 >> "$OUT/timeline_$WL.txt"
rm -rf "$OUT/tl_$WL"
cat "$OUT/timeline_$WL.txt"
