#!/bin/bash
# AddressSanitizer + UBSan over the host layer (loader, BVH builder, PNG decoder) on the CPU.  Host code only: the GPU
# side cannot be sanitized on this pool.  usage: tools/sanitize/run.sh
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${TMPDIR:-/tmp}/rt_host_asan
g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined \
    -I"$ROOT/include" "$ROOT/tools/sanitize/host_asan_main.cpp" "$ROOT/raytracing_amd/csrc/host/rt_host.cpp" \
    -lz -pthread -o "$OUT"
ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 "$OUT" "$ROOT/tests/golden/models"
