#!/usr/bin/env python3
"""Soak of the bucket table that is updated in place (api_grid.h grid_assign): N fused steps of a dam break, and every K steps
gridCount / neighborCount against the oracle's update_grid() on the library's current positions (a rebuild by definition).
   python tools/soak_table.py [n=46] [steps=3000] [every=150]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wcsph_amd import dfsph
from oracle import Oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 46
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
every = int(sys.argv[3]) if len(sys.argv) > 3 else 150
pd = dfsph.init_particle(particleDim=n, neighbor_mode="faithful")
nl = pd.liquid_count
o = Oracle(pd.pos.to_numpy(), nl, 0.025, mode="faithful")   # ONE oracle: the grid box is that of the initial positions, as the library's
dfsph.reset_param()
bad = 0
for s in range(1, steps + 1):
    dfsph.step()
    if s % every == 0:
        pd.hash_grid.update_grid()
        o.pos[...] = pd.pos.to_numpy()
        o.update_grid()
        g_ok = np.array_equal(pd.hash_grid.gridCount, np.array(o.gridCount))
        n_ok = np.array_equal(pd.hash_grid.neighborCount.to_numpy()[:nl], np.array(o.neighborCount)[:nl])
        print("step %5d  t %.3f  iters %s  gridCount %s  neighborCount %s  max bucket %d" % (
            s, dfsph.current_time, (dfsph.vs_iter, dfsph.dv_iter, dfsph.pr_iter), g_ok, n_ok, int(np.max(pd.hash_grid.gridCount))), flush=True)
        bad += (not g_ok) + (not n_ok)
import ctypes as C
from wcsph_amd import _lib
v = C.c_float(); ctx = pd._need_ctx()
_lib.load().wcsph_get_scalar(ctx, _lib.SCALAR_ID["table_updates_in_place"], C.byref(v))
print("updates in place: %d of %d;  mismatches: %d" % (int(v.value), steps + steps // every, bad))
pd.close()
sys.exit(1 if bad else 0)
