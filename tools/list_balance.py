#!/usr/bin/env python3
"""How evenly the compact neighbour lists load the waves, early and late in a C3 run (wcsph_list_stats):
python tools/list_balance.py [late_step] [n]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wcsph_amd import dfsph, _lib
late = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
n = int(sys.argv[2]) if len(sys.argv) > 2 else 126
pd = dfsph.init_particle(particleDim=n, neighbor_mode="faithful")
dfsph.reset_param()
done = 0
for tgt in (10, late):
    while done < tgt:
        dfsph.step(); done += 1
    pd.hash_grid.update_grid()
    h = np.zeros((5, 64), np.int64)
    ctx = pd._need_ctx()
    _lib.check(ctx, _lib.load().wcsph_list_stats(ctx, h.ctypes.data))
    b = np.arange(64)
    npart, nwave = h[0].sum(), h[2].sum()
    useful_l, useful_s = (h[0] * b).sum(), (h[1] * b).sum()
    lanes_l, lanes_s = 64 * (h[2] * b).sum(), 64 * (h[3] * b).sum()
    lanes_m = 64 * (h[4] * b).sum()
    print("step %d: %d particles, %d waves" % (tgt, npart, nwave))
    print("  liquid blocks/particle mean %.2f; hist %s" % (useful_l / npart, {int(k): int(v) for k, v in zip(b, h[0]) if v}))
    print("  solid  blocks/particle mean %.2f; hist %s" % (useful_s / npart, {int(k): int(v) for k, v in zip(b, h[1]) if v}))
    print("  wave max liquid hist %s" % {int(k): int(v) for k, v in zip(b, h[2]) if v})
    print("  wave max solid  hist %s" % {int(k): int(v) for k, v in zip(b, h[3]) if v})
    print("  wave max (liquid + solid) hist %s" % {int(k): int(v) for k, v in zip(b, h[4]) if v})
    print("  lane-trips (lane = particle): liquid %.3g solid %.3g total %.3g; useful %.3g  -> efficiency %.2f" % (
        lanes_l, lanes_s, lanes_l + lanes_s, useful_l + useful_s, (useful_l + useful_s) / (lanes_l + lanes_s)))
    print("  one trip sequence for both streams (device.h walk()): %.3g lane-trips -> efficiency %.2f" % (lanes_m, (useful_l + useful_s) / lanes_m), flush=True)
pd.close()
