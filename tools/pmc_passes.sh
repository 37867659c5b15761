#!/bin/bash
# PMC passes for one kernel group (each pass its own run; --kernel-trace only, per the pool rules).
# Summaries are extracted on the box (the rocpd databases are too big to ship back).
# usage: tools/pmc_passes.sh <group> <kernel-regex> <out.txt> [mode] [steps before the measurement: 5 = at rest, 1200 = developed]
G=${1:-pressure_iter}; RX=${2:-"k_vel_from_k|k_drho_press"}; OUTTXT=${3:-/root/repo/gpurun_out/pmc_summary.txt}; R=/root/repo
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcw; mkdir -p /tmp/pmcw
: > $OUTTXT
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" \
         "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
         "TCP_TOTAL_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES" \
         "TCP_TCP_TA_ADDR_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES" \
         "TCC_HIT TCC_MISS TCC_REQ" \
         "TA_BUSY TA_TOTAL_WAVEFRONTS"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "$RX" -d /tmp/pmcw/p$i -o pmc -- python $R/tools/run_kernel_group.py $G 3 126 ${4:-faithful} ${5:-5} > /tmp/pmcw/p$i.log 2>&1
  echo "## pass $i: $C (rc=$?)" >> $OUTTXT
  for K in $(echo $RX | tr '|' ' '); do
    echo "# $K" >> $OUTTXT
    python $R/tools/rocpd_pmc_kernel.py $K /tmp/pmcw/p$i/*.db >> $OUTTXT 2>&1
  done
done
rm -rf /tmp/pmcw
