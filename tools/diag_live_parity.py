#!/usr/bin/env python3
"""Which particles carry the largest vel_guess / vel error of the live-state parity tests (tests/test_gpu_configs.py), and what do
they have in common?  python tools/diag_live_parity.py [n=46] [squeeze=1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import Oracle
from wcsph_amd import dfsph
from test_gpu_configs import live_state

n = int(sys.argv[1]) if len(sys.argv) > 1 else 46
sq = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
liquid, solid, vel0, omega0 = live_state(n, sq)
pd = dfsph.init_particle(particleDim=n, neighbor_mode="faithful", liquid=liquid, boundary=solid)
nl = pd.liquid_count
o = Oracle(pd.pos.to_numpy(), nl, 0.025, mode="faithful")
dfsph.reset_param(); o.reset_param()
pd.vel.from_numpy(vel0); pd.omega.from_numpy(omega0)
o.vel[...] = vel0; o.omega[...] = omega0
pos0 = pd.pos.to_numpy()[:nl].copy()
dfsph.step(); o.dfsph_step()
for name in ("vel_guess", "vel", "d_vel", "cg_r", "cg_s", "cg_Ad", "cg_dir"):
    a, b = getattr(pd, name).to_numpy(), getattr(o, name)
    d = np.abs(a.astype(np.float64) - b).max(axis=1)
    scale = np.abs(b).max()
    top = np.argsort(-d)[:8]
    print("==", name, "scale %.4g" % scale, "max rel %.3e" % (d.max() / scale), "n(>0.5 max) %d" % int((d > 0.5 * d.max()).sum()))
    for i in top[:5]:
        print("   i=%d err=%.3e gpu=%s orc=%s pos=%s ncount=%d" % (i, d[i], a[i], b[i], pos0[i], o.neighborCount[i]))
print("iters", dfsph.vs_iter, dfsph.dv_iter, dfsph.pr_iter, o.vs_iter, o.dv_iter, o.pr_iter)
print("cg_delta gpu/orc", float(pd.cg_delta[0]) if hasattr(pd, "cg_delta") else None, o.cg_delta, o.cg_delta_zero)
