#!/bin/bash
# HBM-side traffic of one kernel group: FETCH_SIZE and WRITE_SIZE, each in its OWN rocprofv3 --pmc run
# (--kernel-trace only, per the pool rules); summaries are extracted on the box.
# usage: tools/pmc_traffic.sh <group> <kernel-regex> <out.txt> [mode]
G=${1:-pressure_iter}; RX=${2:-"k_vel_from_k|k_drho_press"}; OUTTXT=${3:-/root/repo/gpurun_out/pmc_traffic.txt}; R=/root/repo
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmct; mkdir -p /tmp/pmct
: > $OUTTXT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 180 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "$RX" -d /tmp/pmct/$C -o pmc -- python $R/tools/run_kernel_group.py $G 20 126 ${4:-faithful} > /tmp/pmct/$C.log 2>&1
  echo "## $C (rc=$?)" >> $OUTTXT
  for K in $(echo $RX | tr '|' ' '); do
    echo "# $K" >> $OUTTXT
    python $R/tools/rocpd_pmc_kernel.py $K /tmp/pmct/$C/*.db >> $OUTTXT 2>&1
  done
done
rm -rf /tmp/pmct
