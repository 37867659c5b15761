#!/usr/bin/env python3
"""Stability soak: python tools/long_run.py [config] [steps] [mode]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wcsph_amd import dfsph, scene
n = scene.CONFIGS.get(sys.argv[1] if len(sys.argv) > 1 else "C3")
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
pd = dfsph.init_particle(particleDim=n, neighbor_mode=sys.argv[3] if len(sys.argv) > 3 else "faithful")
dfsph.reset_param()
t0 = time.perf_counter(); it = np.zeros(3); mx = np.zeros(3, int)
for s in range(steps):
    dfsph.step()
    v = np.array([dfsph.vs_iter, dfsph.dv_iter, dfsph.pr_iter]); it += v; mx = np.maximum(mx, v)
    if (s + 1) % 250 == 0:
        p = pd.pos.to_numpy()[:pd.liquid_count]; rho = pd.rho.to_numpy()
        print("step %d t=%.3f dt=%.5f mean iters %s max %s  y-range [%.3f, %.3f]  rho [%.1f, %.1f]  nan=%s  %.2f ms/step" % (
            s + 1, dfsph.current_time, float(dfsph.deltaT[0]), np.round(it / (s + 1), 2), mx, p[:, 1].min(), p[:, 1].max(), rho.min(), rho.max(),
            bool(np.isnan(p).any()), (time.perf_counter() - t0) / (s + 1) * 1e3), flush=True)
pd.close()
