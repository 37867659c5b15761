#!/usr/bin/env python3
"""A periodic ring of ONE slab driven by the REAL migration / ghost-rebuild code (SlabSim._redistribute_resident), not by the
hand-made ghost copies of tools/overlap_ring.py: the simulation object is told it is the middle rank of three (has a left and a
right neighbour), and its neighbour exchange is closed onto itself -- what leaves to the right arrives from the left one period
lower, and vice versa.  Rows really migrate across the seam (the block drifts along x), the resident state / tail classification
/ swap plan / apply path runs every step as on a rank of a real decomposition, at C4's per-rank shape and halo sizes.

   python tools/ring_real.py [nx=32] [ny=252] [nz=252] [steps=20] [planes=callbacks_torchplan,callbacks,native0,native1] [vx=2.0]
planes: callbacks = the communicator behind wcsph_slab_callbacks, native<m> = the library's RCCL plane in halo schedule m;
_torchplan = the swap plan / list assembly in torch (rounds 5-6) instead of inside the library (csrc/api_migrate.h); _hdr = the rows of a
migration over the caller's exchange with a header round trip even on the native plane (instead of wcsph_slab_migrate).
prints one JSON line: ms per step, ms per step spent in redistribute (host + GPU, measured with a device synchronisation on either
side -- subtract it from the step to compare with tools/overlap_ring.py), host waits per step, rows migrated per step; and the
planes' final states compared (by global id)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wcsph_amd import scene, _lib
from wcsph_amd.slab import SlabSim, HipBackend, STATE_W, GHOST_CELLS as G
import ctypes as C

a = sys.argv[1:]
nx = int(a[0]) if len(a) > 0 else 32
ny = int(a[1]) if len(a) > 1 else 252
nz = int(a[2]) if len(a) > 2 else 252
steps = int(a[3]) if len(a) > 3 else 20
planes = (a[4] if len(a) > 4 else "callbacks_torchplan,callbacks,native0,native1").split(",")
vx = float(a[5]) if len(a) > 5 else 2.0
mode = os.environ.get("RING_MODE", "clean")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
R = scene.PARTICLE_RADIUS


def make(plane):
    liquid = scene.dam_break_liquid(nx, ny, nz)
    rng = np.random.default_rng(3)
    liquid = liquid + rng.uniform(-0.2, 0.2, liquid.shape) * R
    L = max(ny, nz) * 2 * R
    solid = scene.box_shell(L)
    bounds = (np.array([-L, 0.0, -L], np.float32), np.array([L, 2 * L, L], np.float32))
    inv = np.float32(1.0 / (2 * R))
    cx = np.trunc((liquid[:, 0].astype(np.float32) - bounds[0][0]) * inv).astype(np.int64)
    lo, hi = int(cx.min()), int(cx.max()) + 1
    sim = SlabSim(lambda **kw: HipBackend(**kw), liquid, solid, bounds, [lo, hi], len(liquid), 0, 1, device=dev, neighbor_mode=mode)
    period = np.float32((hi - lo) * 2 * R)
    nl = len(liquid)
    base = plane.replace("_torchplan", "").replace("_hdr", "").replace("_nosplit", "")
    if base.startswith("native"):
        sim.enable_native_comm(neighbours=(0, 0))
        sim.B.comm_set_mode(int(base[-1]))
    sim.library_migrate, sim.native_migrate = "_torchplan" not in plane, "_hdr" not in plane
    sim.B._chk(sim.B.lib.wcsph_slab_migrate_set_seam(sim.B.ctx, -float(period), float(period), nl))
    # from here on: the middle rank of three, neighbours closed onto itself
    sim.rank, sim.world = 1, 3
    sim.loopback, sim.force_comm = True, True
    sim.interior = (lo + G, hi - G)
    sim.set_overlap(not base.endswith("0") and "_nosplit" not in plane)
    sim._allreduce_tensor = lambda t, op=None: None
    sim._allreduce = lambda v, op: v
    stat = {"red_s": 0.0, "migrated": 0, "n": 0, "ghost_seq": 0}

    def exchange_var(to_left, to_right, head_l=(), head_r=()):
        sim.n_host_waits += 1                                   # the header exchange of the real thing
        from_left, from_right = to_right.clone(), to_left.clone()
        from_left[:, 0] -= float(period); from_right[:, 0] += float(period)
        a_l, a_r = int(head_r[0]), int(head_l[0])                # migrants come first in either message
        # the ghost copies must not share a global id with the row they copy (on a real decomposition they are another rank's rows)
        ng_l, ng_r = from_left.shape[0] - a_l, from_right.shape[0] - a_r
        gid = torch.arange(nl, nl + ng_l + ng_r, dtype=torch.int32, device=dev)
        from_left[a_l:, STATE_W - 1] = gid[:ng_l].view(torch.float32)
        from_right[a_r:, STATE_W - 1] = gid[ng_l:].view(torch.float32)
        return from_left, from_right, list(head_r), list(head_l)
    sim._exchange_var = exchange_var
    orig = sim.redistribute

    def timed_redistribute():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        orig()
        torch.cuda.synchronize(); stat["red_s"] += time.perf_counter() - t0; stat["n"] += 1
    if os.environ.get("RING_NOSYNC") != "1":     # RING_NOSYNC=1: the step as it runs (no synchronisation around the redistribute: the host runs ahead)
        sim.redistribute = timed_redistribute
    sim.rows[:, 3] = vx
    return sim, stat, nl


out = {"shape": [nx, ny, nz], "mode": mode, "steps": steps, "vx": vx}
final = {}
for plane in planes:
    free0 = torch.cuda.mem_get_info()[0]
    sim, stat, nl = make(plane)
    for _ in range(4):
        sim.step()
    stat.update(red_s=0.0, n=0)
    w0, m0 = sim.n_host_waits, sim.n_rows_arrived
    free1 = torch.cuda.mem_get_info()[0]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        sim.step()
        assert sim.n_owned == nl, (sim.n_owned, nl)             # nobody lost on the ring
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / steps
    red = stat["red_s"] / max(stat["n"], 1)
    out[plane] = {"ms_per_step": round(t * 1e3, 3), "redistribute_ms_per_step": round(red * 1e3, 3) if stat["n"] else None,
                  "step_without_redistribute_ms": round((t - red) * 1e3, 3) if stat["n"] else None,
                  "host_waits_per_step": round((sim.n_host_waits - w0) / steps, 2), "rows_migrated_per_step": round((sim.n_rows_arrived - m0) / steps, 1), "migrations_inside_the_library_over_rccl": getattr(sim, "n_native_migrations", 0),
                  "halo_rows": [int(sim.send_l.numel()), int(sim.send_r.numel())], "iters_last": [sim.vs_iter, sim.dv_iter, sim.pr_iter]}
    out[plane]["device_memory_taken_during_the_timed_steps_MB"] = round((free1 - torch.cuda.mem_get_info()[0]) / 1e6, 1)   # (a long run: nothing may grow)
    rows = sim.rows.cpu().numpy()
    gid = rows[:, 14].copy().view(np.int32)
    assert np.array_equal(np.sort(gid), np.arange(nl)), "global ids are no longer a permutation"
    final[plane] = rows[np.argsort(gid)]
    sim.B.close()
ref = final[planes[0]]
vs = float(np.abs(ref[:, 3:6]).max())
for p in planes[1:]:
    d = final[p]
    out[p]["vs_" + planes[0]] = {"pos": float(np.abs(d[:, :3] - ref[:, :3]).max()), "vel_rel": float(np.abs(d[:, 3:6] - ref[:, 3:6]).max() / vs)}
print(json.dumps(out))
