#!/usr/bin/env python3
"""How much does neighbour-list misalignment across lanes cost?  Times the hot kernel groups on the C3 scene after
`k` steps: k = 1 is still the perfect lattice (every lane has the same neighbourhood shape), later steps are disordered."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wcsph_amd import dfsph, _lib
pd = dfsph.init_particle(particleDim=126, neighbor_mode="clean"); dfsph.reset_param()
ctx, lib = pd._need_ctx(), _lib.load()
done = 0
for k in (1, 5, 20, 60, 150):
    while done < k:
        dfsph.step(); done += 1
    out = []
    for name in ("pressure_vel_loop", "pressure_drho_loop", "density_coff", "visc_iter"):
        ms = C.c_float(); _lib.check(ctx, lib.wcsph_time_kernel(ctx, name.encode(), 10, C.byref(ms))); out.append("%s %.1f" % (name, ms.value * 1e3))
    nc = pd.hash_grid.neighborCount.to_numpy()
    print("steps %3d  dt %.4f  iters %s  mean raw count %.1f | %s" % (k, dfsph.deltaT[0], (dfsph.vs_iter, dfsph.dv_iter, dfsph.pr_iter), nc.mean(), "  ".join(out)))
