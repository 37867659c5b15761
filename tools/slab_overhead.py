#!/usr/bin/env python3
"""world = 1 SlabSim (phase-by-phase choreography, state load/store, no comm) vs the fused C path, same scene/mode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wcsph_amd import dfsph, scene
from wcsph_amd.slab import SlabSim, HipBackend, dam_break_slab

n = int(sys.argv[1]) if len(sys.argv) > 1 else 126
steps, warm = 30, 10
liquid, solid, bounds, cuts, ng, gid = dam_break_slab(n, 0, 1)
dev = torch.device("cuda", 0)
sim = SlabSim(lambda **kw: HipBackend(**kw), liquid, solid, bounds, cuts, ng, 0, 1, device=dev)
for _ in range(warm): sim.step()
w0 = getattr(sim, "n_host_waits", 0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): sim.step()
torch.cuda.synchronize(); t_slab = (time.perf_counter() - t0) / steps
waits = (getattr(sim, "n_host_waits", 0) - w0) / steps
sim.B.close()
pd = dfsph.init_particle(particleDim=n, neighbor_mode="clean"); dfsph.reset_param()
for _ in range(warm): dfsph.step()
pd.sync(); t0 = time.perf_counter()
for _ in range(steps): dfsph.step()
pd.sync(); t_fused = (time.perf_counter() - t0) / steps
print("n=%d  slab(world=1) %.3f ms/step   fused(clean) %.3f ms/step   overhead %.3f ms   host waits per step (slab driver) %.1f" % (n, t_slab * 1e3, t_fused * 1e3, (t_slab - t_fused) * 1e3, waits))
