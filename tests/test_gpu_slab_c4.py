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
    from test_slab_gloo import _free_port
    out = str(tmp_path)
    mp.spawn(_rank, args=(world, _free_port(), out, n), nprocs=world, join=True)
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
