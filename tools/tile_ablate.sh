#!/bin/bash
# tile kernels: full / stage-only / compute-only (results are garbage in the ablated runs)
cd /root/repo
for A in 0 1 2; do
  echo "ablate=$A: $(WCSPH_TILES=1 WCSPH_TILE_ABLATE=$A python tools/run_kernel_group.py pressure_vel_loop 10 126 clean 2>&1 | tail -1)  $(WCSPH_TILES=1 WCSPH_TILE_ABLATE=$A python tools/run_kernel_group.py pressure_drho_loop 10 126 clean 2>&1 | tail -1)"
done
