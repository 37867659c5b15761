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
print("iters", dfsph.vs_iter, dfsph.dv_iter, dfsph.pr_iter)
pd.close()
