"""GPU: the library's native RCCL data plane (csrc/api_comm.h: wcsph_comm_init / wcsph_halo_exchange) on ONE GPU.

The pool hands out single-GPU boxes and RCCL refuses two ranks on one GPU, so the data plane is exercised LOOP-BACK: a
communicator of one rank whose left and right neighbour are the rank itself (what is sent to the left arrives from the right,
as on a ring).  The scene is made periodic by hand: the ghost rows of the slab are copies of its own rows within two cell
columns of the opposite end, shifted by the slab's width -- physically a ring of one slab; what matters here is that every halo
refresh of wcsph_slab_step moves real rows and every all-reduce runs, through
  (a) the caller's communicator behind wcsph_slab_callbacks (torch copies, wcsph_amd/slab.py), and
  (b) grouped ncclSend / ncclRecv + ncclAllReduce inside the library,
from the same state: the exchanged buffers bit for bit, the state after three steps to rounding level (new work: the
reference step, dfsph.py:606-617, is single-device)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
pytestmark = pytest.mark.gpu
G = 2   # ghost layer: the walk radius in cell columns


def _ring_sim(mode, n=12):
    import slab_util
    from wcsph_amd.slab import SlabSim, HipBackend, STATE_W
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    liquid, solid, bounds = slab_util.test_scene(n)
    sim = SlabSim(lambda **kw: HipBackend(**kw), liquid, solid, bounds, [0, 10 ** 6], len(liquid), 0, 1, device=dev, neighbor_mode=mode)
    cx = sim._cellx(sim.rows)
    blk_lo, blk_hi = int(cx.min().item()), int(cx.max().item()) + 1
    period = np.float32((blk_hi - blk_lo) * 2 * sim.particleR)
    # a velocity field of the ring's period with a divergence: the divergence solver, the viscosity PCG and the vorticity sums all
    # have work, and the rows at the two ends feel their neighbours across the seam
    sim.rows[:, 3] = 0.5 * torch.sin(2.0 * np.pi * sim.rows[:, 0] / float(period))
    sim.rows[:, 4] = -0.5
    sim.loopback, sim.force_comm, sim.overlap = True, True, True
    sim.interior = (blk_lo + G, blk_hi - G)
    nl = len(liquid)

    def ring_redistribute():
        rows = sim.rows
        cx = sim._cellx(rows)
        near_l = torch.nonzero(cx < blk_lo + G).flatten()
        near_r = torch.nonzero(cx >= blk_hi - G).flatten()
        gl = rows.index_select(0, near_r).clone(); gl[:, 0] -= float(period)      # from the left: the right end, one period down
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
    sim.redistribute = ring_redistribute
    return sim


@pytest.mark.parametrize("mode", ["clean", "faithful"])
def test_native_rccl_data_plane_loopback_equals_the_callback_path(mode):
    out = {}
    # native0 / 1 / 2: wcsph_comm_set_mode -- the exchange inline on the context's stream (passes not split), on the halo stream
    # beside the interior pass, and with the unpack and the boundary pass on the halo stream as well (round 6)
    for plane in ("callbacks", "native0", "native1", "native2"):
        sim = _ring_sim(mode)
        if plane.startswith("native"):
            sim.enable_native_comm(neighbours=(0, 0))
            sim.B.comm_set_mode(int(plane[-1]))
            if plane == "native0":
                sim.overlap = False
        log = []
        for _ in range(3):
            sim.step()
            log.append((sim.vs_iter, sim.dv_iter, sim.pr_iter, float(sim.dt)))
        torch.cuda.synchronize()
        out[plane] = (sim.rows.cpu().numpy().copy(), log, sim.n_exchanges, sim.B.comm_stats() if plane.startswith("native") else None)
        assert sim.send_all.numel() > 100 and sim.ghost_all.numel() > 100          # real rows travelled
        sim.B.close()
    for plane in ("native0", "native1", "native2"):
        _compare_planes(out["callbacks"], out[plane], plane)
    ra = out["callbacks"][0]
    vscale = float(np.abs(ra[:, 3:6]).max())
    # and the ghosts mattered: the same rows without any ghost evolve differently
    sim = _ring_sim(mode)
    sim.loopback = sim.force_comm = sim.overlap = False
    from wcsph_amd.slab import SlabSim
    sim.redistribute = lambda: SlabSim.redistribute(sim)
    for _ in range(3):
        sim.step()
    torch.cuda.synchronize()
    dv = float(np.abs(sim.rows.cpu().numpy()[:, 3:6] - ra[:, 3:6]).max()) / vscale
    print("without the ghosts: velocity differs by %.2e of its scale" % dv)
    assert dv > 2e-2
    sim.B.close()


def _compare_planes(a, b, label):
    (ra, la, xa, _), (rb, lb, xb, st) = a, b
    assert la == lb, (label, la, lb)
    assert xa == xb and xa >= 3 * 9, (xa, xb)                                        # the same number of halo refreshes, >= 9 per step
    assert st[0] == xb and st[1] >= 3 * 4, st                                       # ... issued by the library itself, and its all-reduces
    # The state after three steps.  Two runs of the SAME path already differ in the last bits (the PCG's dot products and the
    # solvers' error sums are double-precision atomics over workgroup partials: their order is the scheduler's), so the two
    # planes are compared at rounding level on the velocity's scale -- three orders of magnitude below what the ghosts change
    # (checked last) -- and bit for bit where nothing is summed: the exchange itself (next test).
    names = ["pos"] * 3 + ["vel"] * 3 + ["omega"] * 3 + ["vel_guess"] * 3 + ["kappa", "kappa_v"]
    vscale = float(np.abs(ra[:, 3:6]).max())
    for c, nm in enumerate(names):
        scale = {"pos": 1.0, "vel": vscale, "vel_guess": vscale}.get(nm, max(float(np.abs(ra[:, c]).max()), 1e-12))
        d = float(np.abs(ra[:, c] - rb[:, c]).max()) / scale
        print("%s vs callbacks %-9s col %2d: max diff / scale = %.2e" % (label, nm, c, d))
        assert d < (5e-7 if nm == "pos" else 5e-6), (nm, c, d)
    np.testing.assert_array_equal(ra[:, 14].view(np.int32), rb[:, 14].view(np.int32))


def test_halo_exchange_and_allreduce_entry_points():
    """wcsph_halo_exchange / wcsph_comm_allreduce on their own: a loop-back exchange swaps the two halves, an all-reduce over one
    rank leaves its buffer unchanged (SUM and MAX, int32 and float64)."""
    import ctypes as C
    from wcsph_amd import _lib
    sim = _ring_sim("clean", n=8)
    sim.enable_native_comm(neighbours=(0, 0))
    lib, ctx = sim.B.lib, sim.B.ctx
    W, nl_, nr_ = 3, 40, 25
    send = torch.arange((nl_ + nr_) * W, dtype=torch.float32, device=sim.device).reshape(-1, W)
    recv = torch.full(((nl_ + nr_), W), -1.0, dtype=torch.float32, device=sim.device)
    with torch.cuda.stream(sim.stream):
        # rows [0, 40) go left and come back from the right (rows [25, 65) of recv); rows [40, 65) go right and come back from the left
        c = sim.B
        idx = torch.zeros(nl_ + nr_, dtype=torch.int32, device=sim.device)
        _lib.check(ctx, lib.wcsph_slab_set_halo(ctx, C.c_void_p(idx.data_ptr()), nl_ + nr_, C.c_void_p(idx.data_ptr()), nl_ + nr_))
        _lib.check(ctx, lib.wcsph_comm_set_halo_split(ctx, nl_, nr_))
        _lib.check(ctx, lib.wcsph_halo_exchange(ctx, W, C.c_void_p(send.data_ptr()), nl_ + nr_, C.c_void_p(recv.data_ptr()), nl_ + nr_))
        words = torch.arange(1000, dtype=torch.int32, device=sim.device)
        dbl0 = np.linspace(0, 1, 64)
        dbl = torch.from_numpy(dbl0).to(sim.device)
        _lib.check(ctx, lib.wcsph_comm_allreduce(ctx, C.c_void_p(words.data_ptr()), 1000, 0))
        _lib.check(ctx, lib.wcsph_comm_allreduce(ctx, C.c_void_p(dbl.data_ptr()), 64, 1))
        _lib.check(ctx, lib.wcsph_comm_allreduce(ctx, C.c_void_p(words.data_ptr()), 1, 2))
        c.sync()
    torch.cuda.synchronize()
    r, s = recv.cpu().numpy(), send.cpu().numpy()
    np.testing.assert_array_equal(r[:nr_], s[nl_:])          # from the left: what went right
    np.testing.assert_array_equal(r[nr_:], s[:nl_])          # from the right: what went left
    np.testing.assert_array_equal(words.cpu().numpy(), np.arange(1000, dtype=np.int32))
    np.testing.assert_array_equal(dbl.cpu().numpy(), dbl0)
    sim.B.close()


def test_native_communicator_survives_a_rebuilt_context():
    """A rebalance of the decomposition closes the rank's context and builds a new one (new cell window): the communicator is
    detached from the old context and attached to the new one (wcsph_comm_detach / _attach, SlabSim._set_cuts).  Ring of one slab:
    two steps, context rebuilt, one more step -- against the same three steps without the rebuild."""
    out = {}
    for rebuild in (False, True):
        sim = _ring_sim("clean")
        sim.enable_native_comm(neighbours=(0, 0))
        for k in range(3):
            if rebuild and k == 2:
                keep = sim.rows.clone()                      # materialised from the library: the truth for the new context
                old_ctx = sim.B.ctx
                sim._set_cuts(list(sim.cuts))                # closes the backend, builds a new one, hands the communicator over
                sim.rows = keep
                assert sim.B.native_comm and sim.B.ctx != old_ctx
                sim.loopback, sim.force_comm, sim.overlap = True, True, True
                cx = sim._cellx(sim.rows)
                sim.interior = (int(cx.min().item()) + G, int(cx.max().item()) + 1 - G)
            sim.step()
        torch.cuda.synchronize()
        out[rebuild] = (sim.rows.cpu().numpy().copy(), sim.B.comm_stats())
        sim.B.close()
    (ra, sa), (rb, sb) = out[False], out[True]
    assert sa == sb and sa[0] >= 27                                  # the same exchanges went through the one communicator
    vscale = float(np.abs(ra[:, 3:6]).max())
    assert np.abs(ra[:, :3] - rb[:, :3]).max() < 5e-7 and np.abs(ra[:, 3:6] - rb[:, 3:6]).max() < 5e-6 * vscale


def test_halo_schedule_tuning_picks_a_candidate_and_the_run_goes_on():
    """SlabSim.tune_halo_schedule (bench.py --gpus N calls it before the warm-up): every candidate runs, the fastest stays on."""
    sim = _ring_sim("clean")
    sim.enable_native_comm(neighbours=(0, 0))
    t = sim.tune_halo_schedule(steps=2)
    assert len(t["candidates_ms_per_step"]) == 3 and t["chosen"] in t["candidates_ms_per_step"], t
    assert t["candidates_ms_per_step"][t["chosen"]] == min(t["candidates_ms_per_step"].values())
    chosen = [c for c in sim.HALO_SCHEDULES if c[0] == t["chosen"]][0]
    assert sim.overlap == chosen[2]
    x0 = sim.n_exchanges
    sim.step()
    torch.cuda.synchronize()
    assert sim.n_exchanges - x0 >= 9 and np.isfinite(sim.rows.cpu().numpy()).all()
    sim.B.close()


def _periodic_ring(plane, n=(12, 14, 14), vx=4.0, mode="clean", library_migrate=True, native_migrate=True):
    """tools/ring_real.py in small: ONE slab that believes it is the middle rank of three, its neighbour exchange closed onto itself
    with a shift of one period -- the REAL migration / ghost rebuild (SlabSim._redistribute_resident: tail classification, swap
    plan, wcsph_state_apply) runs every step and rows really cross the seam"""
    from wcsph_amd import scene
    from wcsph_amd.slab import SlabSim, HipBackend, STATE_W
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    R = scene.PARTICLE_RADIUS
    liquid = scene.dam_break_liquid(*n)
    liquid = liquid + np.random.default_rng(3).uniform(-0.2, 0.2, liquid.shape) * R
    L = max(n) * 2 * R
    solid = scene.box_shell(L)
    bounds = (np.array([-L, 0.0, -L], np.float32), np.array([L, 2 * L, L], np.float32))
    cx = np.trunc((liquid[:, 0].astype(np.float32) - bounds[0][0]) * np.float32(1.0 / (2 * R))).astype(np.int64)
    lo, hi = int(cx.min()), int(cx.max()) + 1
    sim = SlabSim(lambda **kw: HipBackend(**kw), liquid, solid, bounds, [lo, hi], len(liquid), 0, 1, device=dev, neighbor_mode=mode)
    period, nl = np.float32((hi - lo) * 2 * R), len(liquid)
    if plane.startswith("native"):
        sim.enable_native_comm(neighbours=(0, 0))
        sim.B.comm_set_mode(int(plane[-1]))
    sim.rank, sim.world = 1, 3
    sim.loopback, sim.force_comm = True, True
    sim.interior = (lo + G, hi - G)
    sim.set_overlap(plane != "native0")
    sim._allreduce_tensor = lambda t, op=None: None
    sim._allreduce = lambda v, op: v
    sim.migrated = 0

    def exchange_var(to_left, to_right, head_l=(), head_r=()):
        sim.n_host_waits += 1
        from_left, from_right = to_right.clone(), to_left.clone()
        from_left[:, 0] -= float(period); from_right[:, 0] += float(period)
        a_l, a_r = int(head_r[0]), int(head_l[0])
        ng_l, ng_r = from_left.shape[0] - a_l, from_right.shape[0] - a_r
        gid = torch.arange(nl, nl + ng_l + ng_r, dtype=torch.int32, device=dev)
        from_left[a_l:, STATE_W - 1] = gid[:ng_l].view(torch.float32)
        from_right[a_r:, STATE_W - 1] = gid[ng_l:].view(torch.float32)
        sim.migrated += a_l + a_r
        return from_left, from_right, list(head_r), list(head_l)
    sim._exchange_var = exchange_var
    sim.library_migrate, sim.native_migrate = library_migrate, native_migrate
    # the same seam for the migration that runs inside the library over its own communicator (wcsph_slab_migrate)
    sim.B._chk(sim.B.lib.wcsph_slab_migrate_set_seam(sim.B.ctx, -float(period), float(period), nl))
    sim.rows[:, 3] = vx
    return sim, nl


def test_rows_migrate_across_the_seam_of_a_periodic_ring_on_every_plane():
    """The migration path itself: a drifting block on a periodic ring of one slab, driven by the real _redistribute_resident (rows
    cross the seam every step; the classification arrives with the step's last mail).  The torch form of the swap plan (rounds 5-6)
    against the one inside the library (csrc/api_migrate.h) behind the caller's exchange, and against wcsph_slab_migrate on the
    library's own communicator in every halo schedule: every variant keeps every particle exactly once, all agree, and on the native
    plane -- the neighbours' counts travel with the step's last mail -- ONE host wait per step is left."""
    res = {}
    variants = {"torch plan": ("callbacks", False, True), "library plan": ("callbacks", True, True),
                "native0": ("native0", True, True), "native1": ("native1", True, True), "native2": ("native2", True, True),
                "native0, rows over the caller's exchange": ("native0", True, False)}
    for label, (plane, lib_mig, nat_mig) in variants.items():
        sim, nl = _periodic_ring(plane, library_migrate=lib_mig, native_migrate=nat_mig)
        w0 = a0 = None
        for k in range(8):
            sim.step()
            assert sim.n_owned == nl
            if k == 1:
                w0, a0 = sim.n_host_waits, sim.n_rows_arrived if lib_mig else sim.migrated
        torch.cuda.synchronize()
        rows = sim.rows.cpu().numpy()
        gid = rows[:, 14].copy().view(np.int32)
        np.testing.assert_array_equal(np.sort(gid), np.arange(nl))
        res[label] = (rows[np.argsort(gid)], (sim.n_rows_arrived if lib_mig else sim.migrated) - a0, (sim.n_host_waits - w0) / 6.0,
                      getattr(sim, "n_native_migrations", 0))
        sim.B.close()
    ra, mig, waits, _ = res["torch plan"]
    assert mig > 50                                                    # rows did cross the seam
    print("periodic ring: %d rows migrated in 6 steps; host waits per step: %s" % (mig, {k: round(v[2], 2) for k, v in res.items()}))
    assert waits <= 2.5
    vscale = float(np.abs(ra[:, 3:6]).max())
    for label, (rb, mb, wb, nm) in res.items():
        assert mb == mig, label
        assert np.abs(ra[:, :3] - rb[:, :3]).max() < 5e-7 and np.abs(ra[:, 3:6] - rb[:, 3:6]).max() < 5e-6 * vscale, label
        if label in ("native0", "native1", "native2"):
            assert nm >= 6 and wb <= 1.2, (label, nm, wb)             # the step's last mail, nothing else
        else:
            assert nm == 0, label
