#!/bin/bash
R=/root/repo; OUT=${1:-$R/gpurun_out/pmc_build_lists.txt}   # counters of k_build_lists at rest (5 steps) and developed (1200): tools/pmc_build_lists.sh <out.txt>
cd /tmp && export TMPDIR=/tmp
: > $OUT
for ST in 5 1200; do
  i=0
  for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY" \
           "TCP_TOTAL_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES" \
           "TCC_HIT TCC_MISS TCC_REQ"; do
    i=$((i+1)); rm -rf /tmp/pmcw
    timeout 200 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "k_build_lists" -d /tmp/pmcw -o pmc -- python $R/tools/run_kernel_group.py build_lists 3 126 faithful $ST > /tmp/pmcw.log 2>&1
    echo "## steps=$ST pass $i: $C" >> $OUT
    python $R/tools/rocpd_pmc_kernel.py k_build_lists /tmp/pmcw/*.db >> $OUT 2>&1
  done
done
rm -rf /tmp/pmcw
