#!/bin/bash
# On the GPU box: tools/quick.sh for each library under variants/ and each value of one knob.  usage: tools/ab2.sh "<workloads>" "<libs>" NAME v1 v2 ...
WL=$1; LIBS=$2; K=$3; shift 3
for lib in $LIBS; do for v in "$@"; do
  echo "== $lib $K=$v"
  RT_CORE_LIB=$PWD/variants/$lib.so tools/quick.sh "$WL" --set $K=$v || exit 1
done; done
