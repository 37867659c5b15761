#!/bin/bash
# the step boundary of the periodic ring (tools/ring_real.py, C4's per-rank shape) under rocprofv3 --kernel-trace: torch plan vs the
# library's migration behind the caller's exchange vs wcsph_slab_migrate over RCCL
TAG=$1; R=/root/repo; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for V in callbacks_torchplan_nosplit callbacks_nosplit native0_hdr native0; do
  rm -rf /tmp/prof
  RING_NOSYNC=1 rocprofv3 --kernel-trace -d /tmp/prof -o ring -- python $R/tools/ring_real.py 32 252 252 16 $V 2.0 > /dev/null 2> /dev/null
  echo "== $V" >> $O/${TAG}_step_boundary.txt
  python $R/tools/rocpd_step_boundary.py /tmp/prof/ring_results.db >> $O/${TAG}_step_boundary.txt 2>&1
done
rm -rf /tmp/prof
