#!/usr/bin/env python3
"""N plain steps of the C3 dam break (for traces): python tools/run_steps.py [steps] [config] [mode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wcsph_amd import dfsph, scene
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = scene.CONFIGS.get(sys.argv[2] if len(sys.argv) > 2 else "C3")
pd = dfsph.init_particle(particleDim=n, neighbor_mode=sys.argv[3] if len(sys.argv) > 3 else "faithful")
dfsph.reset_param()
for _ in range(steps):
    dfsph.step()
pd.sync()
print("iters", dfsph.vs_iter, dfsph.dv_iter, dfsph.pr_iter)
pd.close()
