#!/usr/bin/env python3
"""A slab of C4's per-rank SHAPE on ONE GPU as a loop-back ring: nx x ny x nz liquid particles (default
32 x 252 x 252 = 2.03 M: the eighth of 252^3 a rank owns at 8 GPUs), cut faces of ny x nz cells, ghost layers of two cell columns
= 2 x 127 k rows -- the halos a rank of the 8-GPU run exchanges with EACH neighbour -- made periodic by hand as in
tests/test_gpu_native_comm.py (the ghosts are the slab's own rows near the opposite end, shifted by the slab's width).  Every halo
refresh of wcsph_slab_step moves real rows through
   callbacks   the caller's communicator behind wcsph_slab_callbacks (torch copies on torch's side stream)
   native      grouped ncclSend / ncclRecv inside the library (csrc/api_comm.h); _m1: the exchange on the halo stream beside the
               interior pass (the default); _m2: exchange, unpack and the pass on the rows next to the cuts on the halo stream beside
               the interior pass; _inline: the exchange on the context's stream, passes not split
with and without the interior / boundary split of the neighbour passes (_nosplit).

   python tools/overlap_ring.py [nx=32] [ny=252] [nz=252] [mode=clean] [steps=20] [variants=callbacks,callbacks_nosplit,native_m1,native_m2,native_m1_nosplit,native_inline]
prints one JSON line (ms per step per variant; the ring's own redistribute -- a full state store + load per step, the price of the
periodic seam -- is measured separately and is the same for every variant).
   WCSPH_RING_TRACE=1 python tools/overlap_ring.py ...   -> only `native`, for rocprofv3 --kernel-trace (tools/rocpd_overlap.py)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wcsph_amd import scene
from wcsph_amd.slab import SlabSim, HipBackend, STATE_W, GHOST_CELLS as G

a = sys.argv[1:]
nx = int(a[0]) if len(a) > 0 else 32
ny = int(a[1]) if len(a) > 1 else 252
nz = int(a[2]) if len(a) > 2 else 252
mode = a[3] if len(a) > 3 else "clean"
steps = int(a[4]) if len(a) > 4 else 20
variants = (a[5] if len(a) > 5 else "callbacks,callbacks_nosplit,native_m1,native_m2,native_m1_nosplit,native_inline").split(",")
trace = os.environ.get("WCSPH_RING_TRACE") == "1"
if trace:
    variants = [os.environ.get("WCSPH_RING_VARIANT", "native_m1")]
warm = 6
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
R = scene.PARTICLE_RADIUS


def ring_scene():
    liquid = scene.dam_break_liquid(nx, ny, nz)
    rng = np.random.default_rng(3)
    liquid = liquid + rng.uniform(-0.2, 0.2, liquid.shape) * R
    L = max(ny, nz) * 2 * R
    solid = scene.box_shell(L)
    bounds = (np.array([-L, 0.0, -L], np.float32), np.array([L, 2 * L, L], np.float32))
    return liquid, solid, bounds


def ring_sim(liquid, solid, bounds):
    inv = np.float32(1.0 / (2 * R))
    cx = np.trunc((liquid[:, 0].astype(np.float32) - bounds[0][0]) * inv).astype(np.int64)
    blk_lo, blk_hi = int(cx.min()), int(cx.max()) + 1
    # one slab [blk_lo, blk_hi): the context's window is the slab + two columns either side, as on a rank of a real decomposition;
    # only the solids of that window (floor, lid and z walls under the slab) are loaded
    cuts = [blk_lo, blk_hi]
    sim = SlabSim(lambda **kw: HipBackend(**kw), liquid, solid, bounds, cuts, len(liquid), 0, 1, device=dev, neighbor_mode=mode)
    period = np.float32((blk_hi - blk_lo) * 2 * R)
    sim.rows[:, 3] = 0.5 * torch.sin(2.0 * np.pi * sim.rows[:, 0] / float(period))
    sim.rows[:, 4] = -0.5
    sim.loopback, sim.force_comm, sim.overlap = True, True, True
    sim.interior = (blk_lo + G, blk_hi - G)
    nl = len(liquid)
    t_red = [0.0, 0]

    def ring_redistribute():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rows = sim.rows
        cx = sim._cellx(rows)
        near_l = torch.nonzero(cx < blk_lo + G).flatten()
        near_r = torch.nonzero(cx >= blk_hi - G).flatten()
        gl = rows.index_select(0, near_r).clone(); gl[:, 0] -= float(period)
        gr = rows.index_select(0, near_l).clone(); gr[:, 0] += float(period)
        gid = torch.arange(nl, nl + gl.shape[0] + gr.shape[0], dtype=torch.int32, device=rows.device)
        gl[:, STATE_W - 1] = gid[:gl.shape[0]].view(torch.float32); gr[:, STATE_W - 1] = gid[gl.shape[0]:].view(torch.float32)
        n = rows.shape[0]
        i32 = torch.int32
        sim.send_l, sim.send_r = near_l.to(i32), near_r.to(i32)
        sim.ghost_l = torch.arange(n, n + gl.shape[0], dtype=i32, device=rows.device)
        sim.ghost_r = torch.arange(n + gl.shape[0], n + gl.shape[0] + gr.shape[0], dtype=i32, device=rows.device)
        sim.send_all = torch.cat([sim.send_l, sim.send_r]); sim.ghost_all = torch.cat([sim.ghost_l, sim.ghost_r])
        sim.n_owned = n
        sim.B.load_state_split(rows, (gl, gr))
        torch.cuda.synchronize(); t_red[0] += time.perf_counter() - t0; t_red[1] += 1
    sim.redistribute = ring_redistribute
    sim.t_red = t_red
    return sim


liquid, solid, bounds = ring_scene()
out = {"shape": [nx, ny, nz], "n_liquid": len(liquid), "mode": mode, "steps": steps}
for v in variants:
    sim = ring_sim(liquid, solid, bounds)
    if v.startswith("native"):
        sim.enable_native_comm(neighbours=(0, 0))
    if v.endswith("nosplit") or v.endswith("inline"):
        sim.overlap = False
    if v.endswith("inline"):
        sim.B.comm_set_mode(0)
    if "_m1" in v:
        sim.B.comm_set_mode(1)
    if "_m2" in v:
        sim.B.comm_set_mode(2)
    for _ in range(warm):
        sim.step()
    sim.t_red[0], sim.t_red[1] = 0.0, 0
    x0 = sim.n_exchanges
    torch.cuda.synchronize(); t0 = time.perf_counter()
    its = []
    for _ in range(steps):
        sim.step(); its.append((sim.vs_iter, sim.dv_iter, sim.pr_iter))
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / steps
    red = sim.t_red[0] / max(sim.t_red[1], 1)
    out[v] = {"ms_per_step": round(t * 1e3, 3), "ring_redistribute_ms": round(red * 1e3, 3), "step_without_redistribute_ms": round((t - red) * 1e3, 3),
              "exchanges_per_step": (sim.n_exchanges - x0) / steps, "halo_rows": [int(sim.send_l.numel()), int(sim.send_r.numel())],
              "iters_last": its[-1]}
    sim.B.close()
print(json.dumps(out))
