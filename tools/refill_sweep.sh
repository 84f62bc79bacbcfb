# usage (on the GPU box): WL="c2 c3" NM="8 16 24 32 48" IDLE="8" bash tools/refill_sweep.sh
# frame time and lane utilisation of the refill kernels against refillNodeMin / refillIdle (closest-hit refill forced on)
for w in ${WL:-c2 c3}; do for idle in ${IDLE:-8}; do for nm in ${NM:-8 16 24 32 48}; do
timeout -k 10 200 python bench.py --workload $w --steps 2 --warmup 2 --no-cpu-baseline --set refill=1 --set refillNodeMin=$nm --set refillIdle=$idle 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; u=r['bounce_lane_utilisation']; print('$w nodeMin=$nm idle=$idle', d['ms_per_step'], r['exclusive_kernel_ms_per_frame'], u['node_step'], u['tri_step'], u['wave_node_steps'], u['wave_tri_steps'], r['nodes_per_ray'])" || exit 1
done; done; done
