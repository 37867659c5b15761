#!/bin/bash
# C5 (1 M particles + adhesion ball, Akinci tension on): in-situ kernel statistics of the reference-literal and the repaired
# compute_tension legs on the current source, and the raw-entry count the literal kernels' algorithmic bytes are stated with:
#   tools/c5_profile.sh <tag>   -> gpurun_out/<tag>_c5_literal.md, <tag>_c5_fixed.md
TAG=$1; R=/root/repo; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for leg in 0 1; do
  name=$([ $leg = 0 ] && echo literal || echo fixed)
  rm -rf /tmp/prof
  rocprofv3 --kernel-trace -d /tmp/prof -o c5 -- python $R/tools/run_c5.py 30 $leg > /tmp/c5_$leg.log 2>&1
  { echo "# C5 ($name compute_tension, tension_fixed = $leg): rocprofv3 --kernel-trace of tools/run_c5.py 30 $leg, second half of the run"; grep "^iters" /tmp/c5_$leg.log;
    python $R/tools/rocpd_stats.py /tmp/prof/c5_results.db 0.5; } > $O/${TAG}_c5_$name.md
done
rm -rf /tmp/prof
