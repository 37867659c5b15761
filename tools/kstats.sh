#!/bin/bash
# in-situ kernel statistics of the C3 dam break at rest (steps 14..45) and, with a second argument, in the developed flow:
#   tools/kstats.sh <tag> [dev]        -> gpurun_out/<tag>_rest_kernel_stats.md (, <tag>_dev_kernel_stats.md)
TAG=$1; R=/root/repo; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
rocprofv3 --kernel-trace -d /tmp/prof -o rest -- python $R/tools/run_steps.py 45 > /dev/null 2>&1
python $R/tools/rocpd_stats.py /tmp/prof/rest_results.db 0.6 > $O/${TAG}_rest_kernel_stats.md
if [ -n "$2" ]; then
  rocprofv3 --kernel-trace -d /tmp/prof -o dev -- python $R/tools/run_steps.py 1240 > /dev/null 2>&1
  python $R/tools/rocpd_stats.py /tmp/prof/dev_results.db 0.985 > $O/${TAG}_dev_kernel_stats.md
fi
rm -rf /tmp/prof
