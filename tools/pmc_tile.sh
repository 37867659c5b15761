#!/bin/bash
# (needs tools/variants/lds_tile_walk_experiment.patch applied: profiles/r05_lds_tile_experiment.md)
# PMC passes for the LDS tile walk's kernels (each pass its own run; --kernel-trace only, per the pool rules).
# usage: tools/pmc_tile.sh <group: tile_vel|tile_drho|tile_build> <kernel-regex> <out.txt> [lib.so] [steps]
G=${1:-tile_drho}; RX=${2:-"k_tile_drho"}; OUTTXT=${3:-/root/repo/gpurun_out/pmc_tile.txt}; R=/root/repo
[ -n "$4" ] && export WCSPH_LIB=$4
export WCSPH_TILE=1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcw; mkdir -p /tmp/pmcw
: > $OUTTXT
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
         "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVES_EQ_64 SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INSTS_GDS" \
         "TCP_TOTAL_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES" \
         "TCC_HIT TCC_MISS TCC_REQ" \
         "TA_BUSY TA_TOTAL_WAVEFRONTS"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "$RX" -d /tmp/pmcw/p$i -o pmc -- python $R/tools/run_kernel_group.py $G 3 126 faithful ${5:-5} > /tmp/pmcw/p$i.log 2>&1
  echo "## pass $i: $C (rc=$?)" >> $OUTTXT
  for K in $(echo $RX | tr '|' ' '); do
    echo "# $K" >> $OUTTXT
    python $R/tools/rocpd_pmc_kernel.py $K /tmp/pmcw/p$i/*.db >> $OUTTXT 2>&1
    [ -s /tmp/pmcw/p$i.log ] && grep -i "error\|invalid\|not supported" /tmp/pmcw/p$i.log | head -3 >> $OUTTXT
  done
done
rm -rf /tmp/pmcw
