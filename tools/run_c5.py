#!/usr/bin/env python3
"""N steps of config C5 (1 M particles + adhesion ball, Akinci tension on) for traces:
python tools/run_c5.py [steps] [tension_fixed 0|1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wcsph_amd import dfsph, scene
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
fixed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
liq, sol = scene.dam_break_with_ball(scene.CONFIGS["C5"])
pd = dfsph.init_particle(particleDim=scene.CONFIGS["C5"], neighbor_mode="faithful", liquid=liq, boundary=sol, tension_coff=0.05, tension_coff_b=0.05, tension_fixed=fixed)
dfsph.reset_param()
for _ in range(steps):
    dfsph.step()
pd.sync()
import numpy as np, time
nc = pd.hash_grid.neighborCount.to_numpy()
t0 = time.perf_counter()
for _ in range(10):
    dfsph.step()
pd.sync()
print("iters", dfsph.vs_iter, dfsph.dv_iter, dfsph.pr_iter, "| liquid %d solid %d | raw list entries per particle (HashGrid.neighborCount): mean %.1f max %d | %.3f ms/step"
      % (pd.liquid_count, pd.solid_count, nc.mean(), nc.max(), (time.perf_counter() - t0) / 10 * 1e3))
pd.close()
