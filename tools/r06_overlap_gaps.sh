#!/bin/bash
# where the native plane loses time against the callback plane on the loop-back ring: idle gaps of the GPU by kernel boundary
TAG=$1; MODE=${2:-clean}; R=/root/repo; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for V in native_m1 native_m2 callbacks native_inline; do
  rm -rf /tmp/prof
  WCSPH_RING_TRACE=1 WCSPH_RING_VARIANT=$V rocprofv3 --kernel-trace -d /tmp/prof -o ring -- python $R/tools/overlap_ring.py 32 252 252 $MODE 12 > /dev/null 2> /dev/null
  python $R/tools/rocpd_ring_steps.py /tmp/prof/ring_results.db > $O/${TAG}_ring_gaps_$V.txt 2>&1
done
rm -rf /tmp/prof
