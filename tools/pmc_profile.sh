#!/bin/bash
# PMC passes for the headline workload (run on the GPU box via gpurun).  Counters are collected in
# their own runs, one small group per pass, never together with --stats/--sys-trace.
# usage: tools/pmc_profile.sh <outdir> [workload]
set -u
OUT=${1:-gpurun_out/pmc}; WL=${2:-cornell-box-800x600x256-d30}
mkdir -p "$OUT"; export TMPDIR=/tmp
CMD="python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --workload $WL"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pass$i" -- $CMD > "$OUT/pass$i.log" 2>&1
  echo "pass $i [$grp] exit $?"
done
