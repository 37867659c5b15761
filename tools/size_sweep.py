#!/usr/bin/env python3
"""Throughput of the fused DFSPH step vs problem size on one GPU: python tools/size_sweep.py [mode] [n ...]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wcsph_amd import dfsph
mode = sys.argv[1] if len(sys.argv) > 1 else "faithful"
sizes = [int(a) for a in sys.argv[2:]] or [30, 46, 64, 80, 100, 126, 160, 200]
rows = []
for n in sizes:
    pd = dfsph.init_particle(particleDim=n, neighbor_mode=mode)
    dfsph.reset_param()
    for _ in range(5):
        dfsph.step()
    pd.sync(); t0 = time.perf_counter()
    steps = 30
    for _ in range(steps):
        dfsph.step()
    pd.sync(); dt = (time.perf_counter() - t0) / steps
    rows.append({"n": n, "liquid": pd.liquid_count, "solid": pd.solid_count, "ms_per_step": round(dt * 1e3, 3),
                 "M_particle_steps_per_s": round(pd.liquid_count / dt / 1e6, 1), "iters": [dfsph.vs_iter, dfsph.dv_iter, dfsph.pr_iter]})
    print(json.dumps(rows[-1]), flush=True)
    pd.close()
