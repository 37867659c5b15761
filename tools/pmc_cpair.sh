#!/bin/bash
# counters of the cluster-window walk beside the gather walk (profiles/r06_cluster_pair.md): tools/pmc_cpair.sh <tag>
TAG=$1; R=/root/repo; OUT=$R/gpurun_out/${TAG}_pmc_cpair.txt
cd /tmp && export TMPDIR=/tmp
: > $OUT
for V in c2_t4 c4_t2; do
  for ST in 25 1200; do
    for GRP in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "TCP_TOTAL_CACHE_ACCESSES" "SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
      rm -rf /tmp/pmcw
      WCSPH_LIB=$R/build_variants/cpair_$V.so timeout 300 rocprofv3 --pmc $GRP --kernel-trace --kernel-include-regex "k_drho_press|k_cpair_drho_press" -d /tmp/pmcw -o pmc -- python $R/tools/cpair_pmc_run.py $ST > /tmp/pmcw.log 2>&1
      echo "## variant=$V steps=$ST counters: $GRP" >> $OUT
      for K in "k_drho_press<3" k_cpair_drho_press; do echo "# $K" >> $OUT; python $R/tools/rocpd_pmc_kernel.py $K /tmp/pmcw/*.db >> $OUT 2>&1; done
    done
  done
done
rm -rf /tmp/pmcw
