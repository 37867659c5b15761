#!/bin/bash
R=/root/repo; OUT=$R/gpurun_out/r05t_pmc_lanes.txt
cd /tmp && export TMPDIR=/tmp
: > $OUT
for ST in 5 1200; do
  for G in pressure_iter divergence_iter_vort visc_iter; do
    rm -rf /tmp/pmcw
    timeout 200 rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-trace --kernel-include-regex "k_vel_from_k|k_drho_press|k_drho_div_iter_vort|k_visc_Ad" -d /tmp/pmcw -o pmc -- python $R/tools/run_kernel_group.py $G 3 126 faithful $ST > /tmp/pmcw.log 2>&1
    echo "## steps=$ST group=$G" >> $OUT
    for K in k_vel_from_k k_drho_press k_drho_div_iter_vort k_visc_Ad; do echo "# $K" >> $OUT; python $R/tools/rocpd_pmc_kernel.py $K /tmp/pmcw/*.db >> $OUT 2>&1; done
  done
done
rm -rf /tmp/pmcw
