#!/usr/bin/env python3
"""Runs one named kernel group of libwcsph_hip.so a few times on the C3 dam break (for rocprofv3 --pmc passes):
   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- python tools/run_kernel_group.py pressure_iter 5 [n] [mode] [steps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wcsph_amd import dfsph, _lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "pressure_iter"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = int(sys.argv[3]) if len(sys.argv) > 3 else 126
mode = sys.argv[4] if len(sys.argv) > 4 else "faithful"
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
pd = dfsph.init_particle(particleDim=n, neighbor_mode=mode)
dfsph.reset_param()
for _ in range(steps):
    dfsph.step()
ms = C.c_float()
ctx = pd._need_ctx()
_lib.check(ctx, _lib.load().wcsph_time_kernel(ctx, name.encode(), reps, C.byref(ms)))
print("%s: %.1f us per launch, n_liquid %d" % (name, ms.value * 1e3, pd.liquid_count))
pd.close()
