#!/usr/bin/env python3
"""Development A/B: evolve the C3 dam break with the product library, then time kernel groups with a variant library on
the SAME state (a variant may compute garbage: it never steps the state).
   python tools/ab_passes.py <variant.so> [steps=1200] [groups...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
var = os.path.abspath(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
groups = sys.argv[3:] or ["pressure_vel_loop", "pressure_drho_loop", "visc_iter", "density_coff", "build_lists"]
from wcsph_amd import dfsph, _lib
pd = dfsph.init_particle(particleDim=126, neighbor_mode="faithful")
dfsph.reset_param()
for _ in range(steps):
    dfsph.step()
pos, vel = pd.pos.to_numpy(), pd.vel.to_numpy()
lib, ctx = _lib.load(), pd._need_ctx()
def time_groups(lib, ctx, tag):
    out = {}
    for g in groups:
        v = C.c_float(); _lib.check(ctx, lib.wcsph_time_kernel(ctx, g.encode(), 10, C.byref(v))); out[g] = round(v.value * 1e3, 1)
    print(tag, out, flush=True)
pd.hash_grid.update_grid()
time_groups(lib, ctx, "product")
pd.close()
# the same state in the variant library
_lib._lib = None; _lib.LIB_PATH = var
pd2 = dfsph.init_particle(particleDim=126, neighbor_mode="faithful")
dfsph.reset_param()
pd2.pos.from_numpy(pos); pd2.vel.from_numpy(vel)
pd2.hash_grid.update_grid()
for _ in range(2):
    dfsph.compute_density()
time_groups(_lib.load(), pd2._need_ctx(), "variant")
pd2.close()
