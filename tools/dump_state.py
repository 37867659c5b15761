#!/usr/bin/env python3
"""Positions of the C3 dam break after N steps, for the offline gather model (tools/gather_model.py):
   python tools/dump_state.py 1200 gpurun_out/state_1200.npz [config]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wcsph_amd import dfsph, scene
steps = int(sys.argv[1]); out = sys.argv[2]
n = scene.CONFIGS.get(sys.argv[3] if len(sys.argv) > 3 else "C3")
pd = dfsph.init_particle(particleDim=n, neighbor_mode="faithful")
dfsph.reset_param()
for _ in range(steps):
    dfsph.step()
pd.sync()
pos = pd.pos.to_numpy()
nl = pd.liquid_count
np.savez_compressed(out, pos=pos[:nl].astype(np.float32), n_solid=pd.solid_count, minb=pd.minboundarynp, maxb=pd.maxboundarynp,
                    bs=pd.hash_grid.blockSize.to_numpy()[0], t=dfsph.current_time, iters=(dfsph.vs_iter, dfsph.dv_iter, dfsph.pr_iter))
print("dumped", out, nl, "t", dfsph.current_time)
pd.close()
