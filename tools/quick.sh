#!/bin/bash
# On the GPU box: a few short bench lines for kernel iteration.  usage: tools/quick.sh "<workloads>" [extra bench args]
# prints per workload: Mrays/s, ms per frame, exclusive kernel ms, bounce-ray lane utilisation
WL=${1:-"c3 c2"}; shift || true
for w in $WL; do
  extra=""; [ "$w" = "c2" ] && extra="--spp ${C2SPP:-256}"
  timeout -k 10 ${QUICK_TIMEOUT:-300} python bench.py --workload $w $extra --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline "$@" > /tmp/q_$w.json 2> /tmp/q_$w.err || { echo "$w FAILED"; tail -5 /tmp/q_$w.err; exit 1; }
  python - "$w" <<'PY'
import json, sys
w = sys.argv[1]
j = json.load(open("/tmp/q_%s.json" % w))
r = j["roofline"]
print("%s  %9.1f Mrays/s  %8.2f ms/frame  excl %s  util %s  choice: %s" % (w, j["value"], j["ms_per_step"], r["exclusive_kernel_ms_per_frame"],
      {k: r["bounce_lane_utilisation"][k] for k in ("node_step", "tri_step")}, j["config"]["traversal_kernels"][:60]))
PY
done
