"""BASELINE.json configs[3] (C4): DFSPH 3D, 252^3 = 16 003 008 liquid + ~4.0 M boundary particles over 8 x-slabs.

The pool hands out one-GPU boxes, so the 8 ranks of the decomposition SHARE cuda:0 (gloo process group, halo buffers and
the bucket histogram staged through the host) -- everything except the wire is the production path: HipBackend on slab
windows of the 505^3-cell grid, interior ranks with two neighbours, 8-way cuts, the global `count`-bucket histogram
(20 M buckets, HashGrid.py:108-114) all-reduced every step, the faithful CFL prefix on global particle ids (Q5), one
forced cut rebalance on the way that really moves rows (the run starts from deliberately skewed cuts).
Reference for the comparison: ONE 16 M-particle context of the plain library (`dfsph.step`, which is oracle-checked at
C3 size in test_gpu_configs.py) -- same iteration counts and dt on every rank, every particle owned exactly once, rows
matched by global id.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
pytestmark = pytest.mark.gpu
STEPS, REBALANCE_AT = 3, 1


def _host_gb():
    try:
        import psutil
        return psutil.virtual_memory().available / 2 ** 30
    except Exception:
        return 0.0


def _skewed(cuts):
    """interior cuts pushed alternately left / right by 3 columns: unequal slabs that the forced rebalance evens out"""
    c = list(cuts)
    for k in range(1, len(c) - 1):
        c[k] += 3 if k % 2 else -3
    return c


def _rank(rank, world, port, outdir, n):
    from wcsph_amd.slab import SlabSim, HipBackend, dam_break_slab
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    _, _, _, cuts0, _, _ = dam_break_slab(n, 0, world)            # the balanced cuts ...
    liquid, solid, bounds, cuts, n_global, gid = dam_break_slab(n, rank, world, cuts=_skewed(cuts0))   # ... skewed
    sim = SlabSim(lambda **kw: HipBackend(**kw), liquid, solid, bounds, cuts, n_global, rank, world, device=dev,
                  comm_device=torch.device("cpu"), neighbor_mode="faithful", gid_local=gid, capacity_factor=1.8)
    del liquid
    log = []
    trace = open(os.path.join(os.environ["C4_TRACE"], "trace_r%d.txt" % rank), "w") if os.environ.get("C4_TRACE") else None
    for k in range(STEPS):
        if trace:
            torch.cuda.synchronize(); trace.write("step %d begins (owned %d)\n" % (k, sim.n_owned if hasattr(sim, "n_owned") else -1)); trace.flush()
        if k == REBALANCE_AT and not os.environ.get("C4_NO_REBALANCE"):
            owned_before = sim.n_owned
            assert sim.rebalance(force=True)
            assert sim.cuts == list(cuts0), (sim.cuts, cuts0)     # the lattice has barely moved: back to the balanced cuts
            if trace:
                torch.cuda.synchronize(); trace.write("rebalanced\n"); trace.flush()
        sim.step()
        log.append((sim.vs_iter, sim.dv_iter, sim.pr_iter, float(sim.dt), sim.n_owned))
    torch.cuda.synchronize()
    if trace:
        trace.write("steps done\n"); trace.flush()
    assert os.environ.get("C4_NO_REBALANCE") or (sim.n_rebalanced == 1 and sim.n_owned != owned_before)
    rows = sim.rows
    with torch.cuda.stream(sim.stream):       # the index list must be written on the stream that reads it
        idx = torch.arange(sim.n_owned, dtype=torch.int32, device=dev)
        rho = sim.B.gather_rows("rho", idx)
    torch.cuda.synchronize()
    out = np.concatenate([rows[:, 0:6].cpu().numpy(), rho.cpu().numpy(), rows[:, 14:15].cpu().numpy()], 1)   # pos vel rho gid
    np.save(os.path.join(outdir, "c4_r%d.npy" % rank), out)
    np.save(os.path.join(outdir, "c4log_r%d.npy" % rank), np.array(log))
    sim.B.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,world", [(252, 8)])
def test_C4_16M_eight_slabs_match_one_context(tmp_path, n, world):
    if _host_gb() < 32:
        pytest.skip("needs ~32 GB of free host memory (8 rank processes + the 16 M-particle reference run)")
    free, total = torch.cuda.mem_get_info(0)
    if free < 80 * 2 ** 30:
        pytest.skip("needs ~80 GB of free device memory (8 slab contexts of ~8 GB each)")
    from test_slab_gloo import spawn_ranks
    out = str(tmp_path)
    spawn_ranks(_rank, world, out, n)
    logs = [np.load(os.path.join(out, "c4log_r%d.npy" % r)) for r in range(world)]
    for r in range(1, world):
        np.testing.assert_array_equal(logs[0][:, :3], logs[r][:, :3])          # (vs, dv, pr) equal on every rank
        np.testing.assert_array_equal(logs[0][:, 3], logs[r][:, 3])            # dt: the same global CFL maximum
    owned = np.array([l[:, 4] for l in logs])
    assert (owned.sum(0) == n ** 3).all()                                      # every particle owned exactly once, every step
    assert n ** 3 == 16003008 or n != 252
    assert (owned[:, 0] != owned[:, -1]).any()                                 # the rebalance moved rows between ranks
    # ONE context of the plain library on the whole scene
    from wcsph_amd import dfsph
    pd = dfsph.init_particle(particleDim=n, neighbor_mode="faithful")
    dfsph.reset_param()
    ref_log = []
    for _ in range(STEPS):
        dt = dfsph.step()
        ref_log.append((dfsph.vs_iter, dfsph.dv_iter, dfsph.pr_iter, dt))
    ref_log = np.array(ref_log)
    np.testing.assert_array_equal(ref_log[:, :3], logs[0][:, :3])
    np.testing.assert_allclose(ref_log[:, 3], logs[0][:, 3], rtol=1e-6)
    nl = pd.liquid_count
    pos, vel, rho = pd.pos.to_numpy()[:nl], pd.vel.to_numpy(), pd.rho.to_numpy()
    pd.close()
    seen = np.zeros(nl, bool)
    worst = np.zeros(3)
    for r in range(world):
        a = np.load(os.path.join(out, "c4_r%d.npy" % r))
        gid = a[:, 7].copy().view(np.int32)
        assert not seen[gid].any()
        seen[gid] = True
        worst = np.maximum(worst, [np.abs(a[:, 0:3] - pos[gid]).max(), np.abs(a[:, 3:6] - vel[gid]).max(),
                                   (np.abs(a[:, 6] - rho[gid]) / rho.max()).max()])
    assert seen.all()
    assert worst[0] < 5e-6, worst          # positions
    assert worst[1] < 1e-3, worst          # velocities
    assert worst[2] < 1e-4, worst          # north star: per-particle density error


# ---- the same decomposition on a LIVE state (round 6): from rest v = 0 and most neighbour sums vanish identically -------------
def _live_fields(pos64, gid, n):
    """tests/test_gpu_configs.py `live_state` at particleDim n, computable per rank: the jitter of a particle is a hash of its
    GLOBAL id (no random stream to keep in step across ranks), the velocity / spin fields are analytic in the jittered position"""
    g = gid.astype(np.uint64)
    def u(k):   # uniform in [-0.2, 0.2) from (gid, component): splitmix-style integer hash
        x = (g * np.uint64(3) + np.uint64(k) + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
        x ^= x >> np.uint64(30); x = (x * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
        x ^= x >> np.uint64(27); x = (x * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
        x ^= x >> np.uint64(31)
        return ((x >> np.uint64(11)).astype(np.float64) / float(1 << 53) - 0.5) * 0.4
    p = pos64 + np.stack([u(0), u(1), u(2)], 1) * 0.025
    p = p.astype(np.float32).astype(np.float64)
    s = 20.0 / n
    x, y, z = (p[:, k] * s for k in range(3))
    vel = 0.5 * np.stack([np.sin(3.0 * y), np.cos(2.0 * x + z) - 0.3, np.sin(x - 2.0 * z)], 1)
    omega = 0.2 * np.stack([np.cos(4.0 * z), np.sin(3.0 * x), np.cos(2.0 * y + x)], 1)
    return p, vel.astype(np.float32), omega.astype(np.float32)


def _rank_live(rank, world, port, outdir, n):
    from wcsph_amd.slab import SlabSim, HipBackend, dam_break_slab
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    liquid, solid, bounds, cuts, n_global, gid = dam_break_slab(n, rank, world)
    liquid, vel0, omega0 = _live_fields(liquid, gid, n)
    sim = SlabSim(lambda **kw: HipBackend(**kw), liquid, solid, bounds, cuts, n_global, rank, world, device=dev,
                  comm_device=torch.device("cpu"), neighbor_mode="faithful", gid_local=gid, capacity_factor=1.8)
    sim.rows[:, 3:6] = torch.from_numpy(vel0).to(dev)
    sim.rows[:, 6:9] = torch.from_numpy(omega0).to(dev)
    del liquid
    log = []
    for k in range(2):
        sim.step()
        log.append((sim.vs_iter, sim.dv_iter, sim.pr_iter, float(sim.dt), sim.n_owned))
    torch.cuda.synchronize()
    rows = sim.rows
    with torch.cuda.stream(sim.stream):
        idx = torch.arange(sim.n_owned, dtype=torch.int32, device=dev)
        rho = sim.B.gather_rows("rho", idx)
    torch.cuda.synchronize()
    out = np.concatenate([rows[:, 0:14].cpu().numpy(), rho.cpu().numpy(), rows[:, 14:15].cpu().numpy()], 1)   # 14 state floats, rho, gid
    np.save(os.path.join(outdir, "c4l_r%d.npy" % rank), out)
    np.save(os.path.join(outdir, "c4llog_r%d.npy" % rank), np.array(log))
    sim.B.close()
    dist.barrier()
    dist.destroy_process_group()


def test_C4_16M_eight_slabs_on_a_live_state_match_one_context(tmp_path):
    """252^3 jittered by +-0.2 R with the smooth velocity / spin field of the reference-text fixtures scaled to the block, 8 slabs
    (interior ranks of ~31 columns: an interior of 27 and two boundary zones each) against ONE 16 M-particle context, two steps:
    every row field and the density.  No oracle at this size (its 8 KiB / particle table would need 131 GB), so the bound is the
    `_live_steps` one on the 0.9999 quantile, and the particles beyond it -- the reference's warm-start discontinuity
    (dfsph.py:421) met in another summation order, tests/test_gpu_configs.py -- are counted: at C3 39 ... 104 of 2 M were measured,
    8 x that is allowed here."""
    n, world = 252, 8
    if _host_gb() < 40:
        pytest.skip("needs ~40 GB of free host memory")
    free, total = torch.cuda.mem_get_info(0)
    if free < 80 * 2 ** 30:
        pytest.skip("needs ~80 GB of free device memory (8 slab contexts of ~8 GB each)")
    from test_slab_gloo import spawn_ranks
    from wcsph_amd import dfsph, scene
    out = str(tmp_path)
    spawn_ranks(_rank_live, world, out, n)
    logs = [np.load(os.path.join(out, "c4llog_r%d.npy" % r)) for r in range(world)]
    for r in range(1, world):
        np.testing.assert_array_equal(logs[0][:, :3], logs[r][:, :3])
        np.testing.assert_array_equal(logs[0][:, 3], logs[r][:, 3])
    assert (np.array([l[:, 4] for l in logs]).sum(0) == n ** 3).all()
    liquid, solid = scene.dam_break_dfsph(n)
    liquid, vel0, omega0 = _live_fields(liquid, np.arange(n ** 3, dtype=np.int64), n)
    pd = dfsph.init_particle(particleDim=n, neighbor_mode="faithful", liquid=liquid, boundary=solid)
    del liquid
    dfsph.reset_param()
    pd.vel.from_numpy(vel0); pd.omega.from_numpy(omega0)
    ref_log = []
    for _ in range(2):
        dt = dfsph.step()
        ref_log.append((dfsph.vs_iter, dfsph.dv_iter, dfsph.pr_iter, dt))
    ref_log = np.array(ref_log)
    np.testing.assert_array_equal(ref_log[:, :3], logs[0][:, :3])
    np.testing.assert_allclose(ref_log[:, 3], logs[0][:, 3], rtol=2e-5)
    nl = pd.liquid_count
    one = {"pos": pd.pos.to_numpy()[:nl], "vel": pd.vel.to_numpy(), "omega": pd.omega.to_numpy(), "vel_guess": pd.vel_guess.to_numpy(),
           "kappa": dfsph.kappa.to_numpy()[:, None], "kappa_v": dfsph.kappa_v.to_numpy()[:, None], "rho": pd.rho.to_numpy()[:, None]}
    pd.close()
    cols = {"pos": slice(0, 3), "vel": slice(3, 6), "omega": slice(6, 9), "vel_guess": slice(9, 12), "kappa": slice(12, 13), "kappa_v": slice(13, 14), "rho": slice(14, 15)}
    vmax = float(np.abs(one["vel"]).max())
    err = {k: [] for k in cols}
    seen = np.zeros(nl, bool)
    for r in range(world):
        a = np.load(os.path.join(out, "c4l_r%d.npy" % r))
        gid = a[:, 15].copy().view(np.int32)
        assert not seen[gid].any()
        seen[gid] = True
        for k, sl in cols.items():
            scale = {"pos": 1.0, "vel_guess": vmax}.get(k, max(float(np.abs(one[k]).max()), 1e-6))
            err[k].append(np.abs(a[:, sl].astype(np.float64) - one[k][gid]).max(axis=1) / scale)
    assert seen.all()
    for k in cols:
        d = np.concatenate(err[k])
        bound = {"pos": 2e-6 * 2, "rho": 5e-5}.get(k, 5e-5 * 2)
        nb = int((d > bound).sum())
        print("C4 live, 8 slabs vs one context: %-9s max %.2e  q9999 %.2e  beyond %.1e: %d of %d" % (k, d.max(), np.quantile(d, 0.9999), bound, nb, len(d)))
        assert np.quantile(d, 0.9999) <= bound and nb <= 8 * 104 and d.max() < 0.1, (k, nb, float(d.max()))
