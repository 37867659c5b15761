#!/bin/bash
# halo / compute overlap of the native data plane on one GPU: timings of the three variants, then a kernel
# trace of `native` analysed by tools/rocpd_overlap.py
#   tools/r06_overlap.sh <tag> [mode=clean]   -> gpurun_out/<tag>_overlap_ring.json, <tag>_overlap.md
TAG=$1; MODE=${2:-clean}; R=/root/repo; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python $R/tools/overlap_ring.py 32 252 252 $MODE 20 callbacks,native_m1 > $O/${TAG}_overlap_ring_check.json 2> $O/${TAG}_overlap_ring.err
rm -rf /tmp/prof
WCSPH_RING_TRACE=1 rocprofv3 --kernel-trace -d /tmp/prof -o ring -- python $R/tools/overlap_ring.py 32 252 252 $MODE 12 > /dev/null 2> $O/${TAG}_overlap_trace.err
python $R/tools/rocpd_overlap.py /tmp/prof/ring_results.db 0.5 > $O/${TAG}_overlap.md 2>> $O/${TAG}_overlap_trace.err
python $R/tools/rocpd_stats.py /tmp/prof/ring_results.db 0.5 > $O/${TAG}_overlap_native_kernel_stats.md
rm -rf /tmp/prof
WCSPH_RING_TRACE=1 WCSPH_RING_VARIANT=callbacks rocprofv3 --kernel-trace -d /tmp/prof -o ring -- python $R/tools/overlap_ring.py 32 252 252 $MODE 12 > /dev/null 2>> $O/${TAG}_overlap_trace.err
python $R/tools/rocpd_stats.py /tmp/prof/ring_results.db 0.5 > $O/${TAG}_overlap_callbacks_kernel_stats.md
rm -rf /tmp/prof
