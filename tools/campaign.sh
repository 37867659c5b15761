#!/bin/bash
# One measurement campaign on the GPU box: default bench line, kernel stats (rest / developed flow), PMC passes.
# usage: tools/campaign.sh <tag>      (outputs under gpurun_out/<tag>_*)
TAG=${1:-r03}; R=/root/repo; O=$R/gpurun_out
cd $R
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python tools/slab_overhead.py > $O/${TAG}_slab_overhead.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
rocprofv3 --kernel-trace -d /tmp/prof -o rest -- python $R/tools/run_steps.py 45 > /dev/null 2>&1
python $R/tools/rocpd_stats.py /tmp/prof/rest_results.db 0.6 > $O/${TAG}_rest_kernel_stats.md
rocprofv3 --kernel-trace -d /tmp/prof -o dev -- python $R/tools/run_steps.py 1240 > /dev/null 2>&1
python $R/tools/rocpd_stats.py /tmp/prof/dev_results.db 0.985 > $O/${TAG}_dev_kernel_stats.md
rm -rf /tmp/prof
bash $R/tools/pmc_traffic.sh pressure_iter "k_vel_from_k|k_drho_press" $O/${TAG}_pmc_traffic.txt
bash $R/tools/pmc_passes.sh pressure_iter "k_vel_from_k|k_drho_press" $O/${TAG}_pmc_pressure_iter.txt
bash $R/tools/pmc_traffic.sh grid_update "k_cell_assign|k_build_lists|k_cellval_occ|k_ncount_box|k_scatter_liquid|k_scan_final" $O/${TAG}_pmc_grid_traffic.txt
