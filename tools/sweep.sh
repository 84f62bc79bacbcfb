#!/bin/bash
# Tuning sweep on the GPU box: tools/sweep.sh <outfile> <common bench args> -- <variant args>;<variant args>;...
OUT=$1; shift
COMMON=()
while [ "$1" != "--" ]; do COMMON+=("$1"); shift; done
shift
IFS=';' read -ra VARS <<< "$*"
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for v in "${VARS[@]}"; do
  echo "== $v" >> "$OUT"
  timeout -k 10 300 python bench.py "${COMMON[@]}" $v --no-cpu-baseline 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    line=line.strip()
    if line.startswith('{'):
        j=json.loads(line); r=j['roofline']
        print('value %.1f Mrays/s  ms/step %.2f  kernels %s  launch_ms %.4f frac %.3f' % (j['value'], j['ms_per_step'], r['kernel_ms_per_step'], r['avg_launch_ms'], r['frac']))
    elif line: print(line)
" >> "$OUT" || { echo "FAILED rc=$?" >> "$OUT"; exit 1; }
done
