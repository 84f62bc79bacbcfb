for v in "--set dynamicFetch=0" "--set refillIdle=64" "--set refillIdle=32" "--set refillIdle=8"; do echo "== $v"; timeout -k 10 120 python bench.py --spp 64 --steps 1 --warmup 0 --no-cpu-baseline $v 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.readline()); r=j['roofline']; print(r['bounce_lane_utilisation'], r['nodes_per_ray'], r['tris_per_ray'], r['kernel_ms_per_step'])" || exit 1; done
