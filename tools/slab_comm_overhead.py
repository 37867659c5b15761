#!/usr/bin/env python3
"""What the slab driver costs on ONE GPU, C3 load: ms per step of
   fused       wcsph_dfsph_step (the 1-GPU product path)
   slab        SlabSim world = 1: redistribute (state load) + wcsph_slab_step without communicator + state store
   slab+cb     ... with the communicator called at every exchange point (flags bit 3): Python callbacks, empty halos, the host cost
               of ~20 callbacks per step
   slab+native ... the library's RCCL data plane instead (all-reduces over one rank are real RCCL calls, exchanges are empty)
and the pieces of the driver's own overhead (state store / load replayed alone).
   python tools/slab_comm_overhead.py [n=126] [mode=clean]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wcsph_amd import dfsph
from wcsph_amd.slab import SlabSim, HipBackend, dam_break_slab

n = int(sys.argv[1]) if len(sys.argv) > 1 else 126
mode = sys.argv[2] if len(sys.argv) > 2 else "clean"
steps, warm = 30, 10
dev = torch.device("cuda", 0)
out = {"n": n, "mode": mode}


def run(label, force, native):
    liquid, solid, bounds, cuts, ng, gid = dam_break_slab(n, 0, 1)
    sim = SlabSim(lambda **kw: HipBackend(**kw), liquid, solid, bounds, cuts, ng, 0, 1, device=dev, neighbor_mode=mode, gid_local=gid)
    sim.force_comm = force
    if native:
        sim.enable_native_comm()
    for _ in range(warm): sim.step()
    w0, x0 = sim.n_host_waits, sim.n_exchanges
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): sim.step()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / steps
    out[label] = round(t * 1e3, 3)
    out[label + "_waits"] = (sim.n_host_waits - w0) / steps
    out[label + "_exchange_points"] = (sim.n_exchanges - x0) / steps
    if label == "slab":   # the driver's own pieces, replayed alone
        with torch.cuda.stream(sim.stream):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): rows = sim.B.store_state()
            torch.cuda.synchronize(); out["store_state_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
            t0 = time.perf_counter()
            for _ in range(20): sim.B.load_state(rows.shape[0], rows)
            torch.cuda.synchronize(); out["load_state_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
    if native:
        out["native_comm_stats"] = sim.B.comm_stats()
    sim.B.close()


run("slab", False, False)
run("slab_callbacks", True, False)
run("slab_native", True, True)
pd = dfsph.init_particle(particleDim=n, neighbor_mode=mode); dfsph.reset_param()
for _ in range(warm): dfsph.step()
pd.sync(); t0 = time.perf_counter()
for _ in range(steps): dfsph.step()
pd.sync(); out["fused"] = round((time.perf_counter() - t0) / steps * 1e3, 3)
pd.close()
for k in ("slab", "slab_callbacks", "slab_native"):
    out[k + "_over_fused"] = round(out[k] - out["fused"], 3)
print(json.dumps(out))
