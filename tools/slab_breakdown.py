#!/usr/bin/env python3
"""Where a SlabSim step spends its time (world = 1, real HIP backend): wall time per stage with a device sync after
each, and the same step without the syncs.   python tools/slab_breakdown.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wcsph_amd.slab import SlabSim, HipBackend, dam_break_slab

n = int(sys.argv[1]) if len(sys.argv) > 1 else 126
liquid, solid, bounds, cuts, ng, gid = dam_break_slab(n, 0, 1)
dev = torch.device("cuda", 0)
sim = SlabSim(lambda **kw: HipBackend(**kw), liquid, solid, bounds, cuts, ng, 0, 1, device=dev)
for _ in range(10):
    sim.step()
torch.cuda.synchronize()
B = sim.B
acc = {}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w
orig = dict(redistribute=sim.redistribute, grid_update=B.grid_update, store_state=B.store_state, phase=B.phase, get_acc=B.get_acc, set_acc=B.set_acc)
sim.redistribute = timed("redistribute(load_state)", orig["redistribute"])
B.grid_update = timed("grid_update", orig["grid_update"])
B.store_state = timed("store_state", orig["store_state"])
B.get_acc = timed("get_acc", orig["get_acc"])
B.set_acc = timed("set_acc", orig["set_acc"])
def phase(name, _p=orig["phase"]):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = _p(name)
    torch.cuda.synchronize(); acc["ph:" + name] = acc.get("ph:" + name, 0.0) + time.perf_counter() - t0
    return r
B.phase = phase
steps = 20
t0 = time.perf_counter()
for _ in range(steps):
    sim.step()
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / steps
print("synced step %.3f ms" % (tot * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-28s %8.1f us" % (k, v / steps * 1e6))
print("  sum %.1f us" % (sum(acc.values()) / steps * 1e6))
