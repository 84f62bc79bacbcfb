#!/bin/bash
# On the GPU box: the benchmark workloads, one line each (frame ms, Mrays/s, exclusive kernel ms).  usage: tools/r3_quick.sh <outdir> <tag> [workloads...]
OUT=$1; TAG=$2; shift 2
mkdir -p "$OUT"
for WL in "${@:-c2}"; do
  timeout -k 10 300 python bench.py --workload $WL --no-cpu-baseline ${BENCH_EXTRA:-} > "$OUT/${TAG}_$WL.json" 2> "$OUT/${TAG}_$WL.err" || { echo "$WL failed"; tail -3 "$OUT/${TAG}_$WL.err"; exit 1; }
  python3 -c "
import json,sys
j=json.load(open('$OUT/${TAG}_$WL.json'))
print('$TAG $WL', j['ms_per_step'], 'ms', j['value'], 'Mrays/s', j['roofline']['exclusive_kernel_ms_per_frame'], j['config']['traversal_kernels'][:40])"
done
