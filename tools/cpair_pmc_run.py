#!/usr/bin/env python3
"""(needs tools/variants/cluster_pair_walk_experiment.patch applied; WCSPH_LIB = a -DWCSPH_CPAIR build) `steps` steps of C3, then the density-change pass of pressure_iter three times through the
product's gather walk and three times through the cluster-window walk (csrc/cpair.h), for rocprofv3 --pmc passes:
   python tools/cpair_pmc_run.py [steps=1200]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wcsph_amd import dfsph, _lib
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
lib = _lib.load()
pd = dfsph.init_particle(particleDim=126, neighbor_mode="faithful"); dfsph.reset_param()
ctx = pd._need_ctx()
for _ in range(steps):
    dfsph.step()
pd.hash_grid.update_grid(); dfsph.compute_density(); dfsph.compute_dfsph_coff()
_lib.check(ctx, lib.wcsph_dfsph_begin_pressure(ctx))
ms = C.c_float()
for name, reps in (("pressure_iter", 1), ("drho_press_rec", 2), ("cpair_build", 1), ("cpair_drho", 2)):
    _lib.check(ctx, lib.wcsph_time_kernel(ctx, name.encode(), reps, C.byref(ms)))
    print(name, round(ms.value * 1e3, 1))
pd.close()
