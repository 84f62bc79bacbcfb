#!/bin/bash
# Builds the two micro-benchmarks from their sources on the box and runs them (binaries are never committed).
# usage (GPU box): tools/ubench/run.sh <outdir>    -> <outdir>/ubench_gather_rate.txt, <outdir>/ubench_pk_rate.txt
set -eu
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=${1:-gpurun_out}; mkdir -p "$OUT"
B=${TMPDIR:-/tmp}/rt_ubench; mkdir -p "$B"
for k in gather_rate pk_rate; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -o "$B/$k" "$HERE/$k.hip"
    timeout -k 10 120 "$B/$k" > "$OUT/ubench_$k.txt"
done
tail -n 40 "$OUT/ubench_gather_rate.txt"
