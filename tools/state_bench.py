#!/usr/bin/env python3
"""Step time and kernel-group times of the C3 dam break at several points of its evolution (at rest ... splashed):
   python tools/state_bench.py [config] [mode] [checkpoints, e.g. 25,600,1200]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from wcsph_amd import dfsph, scene, _lib  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
mode = sys.argv[2] if len(sys.argv) > 2 else "faithful"
marks = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "25,600,1200").split(",")]
pd = dfsph.init_particle(particleDim=scene.CONFIGS.get(cfg) or int(cfg), neighbor_mode=mode)
dfsph.reset_param()
lib, ctx = _lib.load(), pd._need_ctx()
done = 0
for m in marks:
    while done < m:
        dfsph.step(); done += 1
    pd.sync()
    t0 = time.perf_counter(); its = []
    for _ in range(20):
        dfsph.step(); done += 1
        its.append((dfsph.vs_iter, dfsph.dv_iter, dfsph.pr_iter))
    pd.sync()
    ms_step = (time.perf_counter() - t0) / 20 * 1e3
    row = {"step": done, "t": round(dfsph.current_time, 4), "ms_per_step": round(ms_step, 3), "iters": np.round(np.mean(its, 0), 2).tolist()}
    for name in ("pressure_iter", "pressure_vel_loop", "pressure_drho_loop", "density_coff", "grid_update", "build_lists", "visc_iter", "visc_minv_residual_vort"):
        v = C.c_float()
        _lib.check(ctx, lib.wcsph_time_kernel(ctx, name.encode(), 10, C.byref(v)))
        row[name + "_us"] = round(v.value * 1e3, 1)
    n = pd.liquid_count
    idx = np.linspace(0, n - 4096, 16).astype(int)
    cl, cs, hp = [], [], []
    for i in idx:
        c, _ = pd.hash_grid.neighbor.compact_rows(int(i), 4096, cap=8)
        cl.append(c[:, 0]); cs.append(c[:, 1]); hp.append(c[:, 2])
    cl, cs, hp = np.concatenate(cl), np.concatenate(cs), np.concatenate(hp)
    row.update(liquid_entries=round(float(cl.mean()), 2), solid_entries=round(float(cs.mean()), 2), hops_mean=round(float(hp.mean()), 3), hops_max=int(hp.max()))
    print(json.dumps(row), flush=True)
pd.close()
