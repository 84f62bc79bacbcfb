#!/bin/bash
# On the GPU box: tools/quick.sh once per value of one tuning knob.  usage: tools/knob_ab.sh "<workloads>" NAME v1 v2 ...
WL=$1; K=$2; shift 2
for v in "$@"; do
  echo "== $K=$v"
  tools/quick.sh "$WL" --set $K=$v || exit 1
done
