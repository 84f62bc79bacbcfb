#!/bin/bash
# On the GPU box: tools/quick.sh for each experimental library under variants/.  usage: tools/ab.sh "<workloads>" lib1 lib2 ...
WL=$1; shift
for v in "$@"; do
  echo "== $v"
  RT_CORE_LIB=$PWD/variants/$v.so tools/quick.sh "$WL" || exit 1
done
