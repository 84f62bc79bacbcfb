#!/bin/bash
# Collects rocprofv3 PMC counters for the bench kernels, one pass per run (gfx950 slot limits: 8 SQ, 4 TCC, 4 TCP,
# 2 TA, 2 GRBM per pass).  Usage on the GPU box: tools/pmc_passes.sh <outdir> [bench args...]
# Never combined with --kernel-trace/--stats or any trace domain.
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
 "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum"
 "FETCH_SIZE"
 "WRITE_SIZE GRBM_GUI_ACTIVE"
 "TA_TA_BUSY_sum TA_BUSY_avr GRBM_TA_BUSY"
 "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_VMEM_TA_ADDR_FIFO_FULL"
)
# PMC_ONLY="1 2": run only these passes
i=0
for p in "${PASSES[@]}"; do
  i=$((i+1))
  if [ -n "${PMC_ONLY:-}" ] && ! [[ " $PMC_ONLY " == *" $i "* ]]; then continue; fi
  timeout -s KILL 240 rocprofv3 --pmc $p --output-format csv -d "$OUT/pass$i" -- python "$ROOT/bench.py" "$@" > "$OUT/pass$i.log" 2>&1
  rc=$?
  echo "pass $i ($p) rc=$rc"
  if [ $rc -ne 0 ]; then tail -3 "$OUT/pass$i.log"; exit $rc; fi
done
