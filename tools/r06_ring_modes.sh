cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
python $R/tools/overlap_ring.py 32 252 252 clean 20 > $O/r06d_overlap_ring.json 2>/dev/null; cat $O/r06d_overlap_ring.json
echo; echo "== memcpy diagnostic (dev build)"
WCSPH_LIB=$R/build_variants/dev.so WCSPH_COMM_LOOPBACK_MEMCPY=1 python $R/tools/overlap_ring.py 32 252 252 clean 20 native_m1,native_m2 > $O/r06d_overlap_ring_memcpy.json 2>/dev/null; cat $O/r06d_overlap_ring_memcpy.json
rm -rf /tmp/prof
WCSPH_LIB=$R/build_variants/dev.so WCSPH_COMM_LOOPBACK_MEMCPY=1 WCSPH_RING_TRACE=1 rocprofv3 --kernel-trace -d /tmp/prof -o ring -- python $R/tools/overlap_ring.py 32 252 252 clean 12 > /dev/null 2>&1
python $R/tools/rocpd_ring_steps.py /tmp/prof/ring_results.db | head -14
