#!/usr/bin/env python3
"""Compact neighbour-list statistics early and late in a C3 run: python tools/list_stats.py <late_step> [mode]"""
import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from wcsph_amd import dfsph, scene
pd = dfsph.init_particle(particleDim=126, neighbor_mode=sys.argv[2] if len(sys.argv) > 2 else "faithful")
dfsph.reset_param()
for tgt in [10, int(sys.argv[1])]:
    while getattr(pd, "_n", 0) < tgt:
        dfsph.step(); pd._n = getattr(pd, "_n", 0) + 1
    pd.hash_grid.update_grid()
    n = pd.liquid_count
    idx = np.linspace(0, n - 4096, 64).astype(int)
    cl = []; cs = []; hp = []
    for i in idx:
        c, l = pd.hash_grid.neighbor.compact_rows(int(i), 4096, cap=8)
        cl.append(c[:, 0]); cs.append(c[:, 1]); hp.append(c[:, 2])
    cl, cs, hp = np.concatenate(cl), np.concatenate(cs), np.concatenate(hp)
    raw = pd.hash_grid.neighborCount.to_numpy()
    print("step %d: liquid entries mean %.1f p99 %d max %d | solid entries mean %.1f max %d | hops mean %.2f max %d | raw list mean %.1f max %d" % (
        tgt, cl.mean(), np.percentile(cl, 99), cl.max(), cs.mean(), cs.max(), hp.mean(), hp.max(), raw.mean(), raw.max()), flush=True)
pd.close()
