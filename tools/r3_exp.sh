#!/bin/bash
# On the GPU box: A/B of experimental library builds (exp/<name>.so, built by `python -m raytracing_amd.build exp/<name>.so -D...`)
# against the product build.  usage: tools/r3_exp.sh <outdir> "<workloads>" <name|base>[:bench args] ...
OUT=$1; WLS=$2; shift 2
for spec in "$@"; do
  v=${spec%%:*}; extra=""; [ "$spec" != "$v" ] && extra=${spec#*:}
  tag=$(echo "$spec" | tr -c 'A-Za-z0-9_\n' '_')
  if [ "$v" = base ]; then unset RT_CORE_LIB; else export RT_CORE_LIB=$PWD/exp/$v.so; fi
  BENCH_EXTRA="$extra" tools/r3_quick.sh "$OUT" "$tag" $WLS || exit 1
done
