#!/usr/bin/env python3
"""Which resource do the two neighbour passes of pressure_iter wait for?  Times the ablation kernels of
csrc/ablate.h (one ingredient removed at a time) on the C3 dam break after a few steps.
    python tools/ablate.py [--config C3] [--steps 10] [--reps 20]"""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--mode", default="faithful")
ap.add_argument("--groups", default="")
a = ap.parse_args()
g.build()
from wcsph_amd import dfsph, scene, _lib
n = scene.CONFIGS.get(a.config) or int(a.config)
pd = dfsph.init_particle(particleDim=n, neighbor_mode=a.mode)
dfsph.reset_param()
for _ in range(a.steps):
    dfsph.step()
pd.hash_grid.update_grid()
ctx, lib = pd._need_ctx(), _lib.load()
names = a.groups.split(",") if a.groups else (["pressure_vel_loop", "pressure_drho_loop"] +
         ["ablate_vel:%d" % k for k in range(4)] + ["ablate_drho:%d" % k for k in range(5)])
out = {}
for nm in names:
    ms = C.c_float()
    _lib.check(ctx, lib.wcsph_time_kernel(ctx, nm.encode(), a.reps, C.byref(ms)))
    out[nm] = round(ms.value * 1e3, 1)
    print("%-24s %8.1f us" % (nm, out[nm]), flush=True)
print(json.dumps(out))
