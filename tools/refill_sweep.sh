# usage (on the GPU box): bash tools/refill_sweep.sh  -- frame time against refillBlocksPerCU with the library's own kernel choice
for w in ${WL:-c3 c5 c4 c2}; do
for bpc in ${BPC:-4 6 8 12 16}; do
timeout -k 10 200 python bench.py --workload $w --steps 2 --warmup 5 --no-cpu-baseline --set refillBlocksPerCU=$bpc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$w bpc=$bpc', d['ms_per_step'], r['exclusive_kernel_ms_per_frame'])" || exit 1
done; done
