#!/usr/bin/env python3
"""Prototype measurement (library built with -DWCSPH_BOX_PROTO, WCSPH_LIB=...): the pressure passes with the block's
neighbourhood staged in LDS against the product's gather walk, and their agreement.  python tools/box_proto_bench.py 25,1200"""
import ctypes as C, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wcsph_amd import dfsph, _lib
marks = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "25,1200").split(",")]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 126
lib = _lib.load()

def run_to(m):
    pd = dfsph.init_particle(particleDim=n, neighbor_mode="faithful"); dfsph.reset_param()
    for _ in range(m): dfsph.step()
    pd.hash_grid.update_grid(); dfsph.compute_density()
    return pd

def t(ctx, name, reps=10):
    v = C.c_float(); _lib.check(ctx, lib.wcsph_time_kernel(ctx, name.encode(), reps, C.byref(v))); return round(v.value * 1e3, 1)

for m in marks:
    # agreement: the same state, two iterations through either path
    out = {}
    for path in ("pressure_iter", "pressure_iter_box"):
        pd = run_to(m); ctx = pd._need_ctx()
        t(ctx, path, 1)
        out[path] = (pd.vel.to_numpy(), pd.adv_rho.to_numpy())
        if path == "pressure_iter_box":
            row = {"step": m}
            for g in ("pressure_iter", "pressure_vel_loop", "pressure_drho_loop", "pressure_iter_box", "pressure_vel_box", "pressure_drho_box"):
                row[g] = t(ctx, g)
        pd.close()
    (va, aa), (vb, ab) = out["pressure_iter"], out["pressure_iter_box"]
    row["vel_rel_diff"] = float(np.abs(va - vb).max() / np.abs(va).max())
    row["adv_rho_rel_diff"] = float(np.abs(aa - ab).max() / np.abs(aa).max())
    print(json.dumps(row), flush=True)
