#!/usr/bin/env python3
"""The grid update's kernels one after the other on ONE stream (the slab halves begin / mid / end: no side stream), for a
kernel trace that shows what each of them costs ALONE:
   rocprofv3 --kernel-trace -d /tmp/prof -o g -- python tools/grid_serial_stats.py [steps_before] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wcsph_amd import dfsph, _lib  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
pd = dfsph.init_particle(particleDim=126, neighbor_mode="faithful")
dfsph.reset_param()
for _ in range(steps):
    dfsph.step()
ctx = pd._need_ctx()
L = _lib.load()
for _ in range(reps):
    _lib.check(ctx, L.wcsph_grid_update_begin(ctx))
    _lib.check(ctx, L.wcsph_grid_update_mid(ctx))
    _lib.check(ctx, L.wcsph_grid_update_end(ctx))
pd.sync()
print("done")
pd.close()
