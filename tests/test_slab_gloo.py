"""N > 1 path on CPU: two gloo ranks run wcsph_amd.slab.SlabSim (partition, migration, ghost rebuild,
halo refreshes between phases, scalar all-reduces) with the checker backend and must reproduce the
single-rank run of the same scene; the single-rank SlabSim must in turn reproduce oracle.dfsph_step()."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

STEPS = 6
VX = 4.0   # initial x velocity: particles cross the cut within a few steps => exercises migration


def _free_port():
    """a port for the rendezvous store, taken BELOW the kernel's ephemeral range (32768-60999): a port from that range can be handed
    out again -- as the source port of some rank's own outgoing connection -- between this probe and the store's listen()
    (seen once in ~600 spawns on the GPU box: EADDRINUSE)"""
    import random
    for _ in range(200):
        p = random.randint(20000, 29999)
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", p))
        except OSError:
            continue
        finally:
            s.close()
        return p
    raise RuntimeError("no free port in 20000-29999")


def spawn_ranks(fn, world, *args):
    """mp.spawn(fn, (world, port) + args) on a fresh port; a rendezvous that loses its port all the same is tried again"""
    for attempt in range(3):
        try:
            return mp.spawn(fn, args=(world, _free_port()) + tuple(args), nprocs=world, join=True)
        except Exception as e:      # noqa: BLE001 -- torch.multiprocessing.ProcessRaisedException carries the child's text
            if attempt == 2 or "address already in use" not in str(e).lower():
                raise


def _run(rank, world, port, outdir, n):
    import slab_util
    from wcsph_amd.slab import SlabSim
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    liquid, solid, bounds = slab_util.test_scene(n)
    cuts, parts = slab_util.split_scene(liquid, bounds, world)
    sim = SlabSim(lambda **kw: slab_util.OracleBackend(**kw), parts[rank], solid, bounds, cuts, len(liquid), rank, world,
                  gid_local=slab_util.split_scene.gids[rank])
    sim.rows[:, 3] = VX
    log = []
    for k in range(STEPS):
        if world > 1 and k == int(os.environ.get("SLAB_TEST_REBALANCE_AT", "-1")):
            assert sim.rebalance(force=True)
        sim.step()
        log.append((sim.vs_iter, sim.dv_iter, sim.pr_iter, float(sim.dt), sim.n_owned))
    np.save(os.path.join(outdir, "rows_w%d_r%d.npy" % (world, rank)), sim.rows.numpy())
    np.save(os.path.join(outdir, "log_w%d_r%d.npy" % (world, rank)), np.array(log))
    sim.B.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_ranks_match_one_rank(tmp_path):
    import slab_util
    n = 8
    out = str(tmp_path)
    _run(0, 1, 0, out, n)
    spawn_ranks(_run, 2, out, n)
    one = np.load(os.path.join(out, "rows_w1_r0.npy"))
    two = np.concatenate([np.load(os.path.join(out, "rows_w2_r%d.npy" % r)) for r in range(2)])
    assert len(one) == len(two) == n ** 3
    dpos, dvel, _ = slab_util.match_rows(one, two)
    assert dpos < 2e-6 and dvel < 2e-4
    l1 = np.load(os.path.join(out, "log_w1_r0.npy"))
    l2 = [np.load(os.path.join(out, "log_w2_r%d.npy" % r)) for r in range(2)]
    np.testing.assert_array_equal(l1[:, :3], l2[0][:, :3])          # same iteration counts on every rank
    np.testing.assert_array_equal(l2[0][:, :3], l2[1][:, :3])
    np.testing.assert_allclose(l1[:, 3], l2[0][:, 3], rtol=1e-6)    # same dt
    assert (l2[0][:, 4] + l2[1][:, 4] == n ** 3).all()              # nobody lost or duplicated
    assert l2[0][0, 4] != l2[0][-1, 4]                              # and ownership really migrated


@pytest.mark.parametrize("world", [2, 3])
def test_rebalanced_cuts_change_nothing(tmp_path, world):
    """SlabSim.rebalance(): new cuts from the all-reduced column histogram, rows routed to their new owners, backends
    rebuilt -- mid-run, after the block has started to move; the trajectory is the 1-rank one."""
    import slab_util
    n = 10
    out = str(tmp_path)
    _run(0, 1, 0, out, n)
    os.environ["SLAB_TEST_REBALANCE_AT"] = "3"
    try:
        spawn_ranks(_run, world, out, n)
    finally:
        os.environ.pop("SLAB_TEST_REBALANCE_AT", None)
    one = np.load(os.path.join(out, "rows_w1_r0.npy"))
    many = np.concatenate([np.load(os.path.join(out, "rows_w%d_r%d.npy" % (world, r))) for r in range(world)])
    assert len(one) == len(many) == n ** 3
    dpos, dvel, _ = slab_util.match_rows(one, many)
    assert dpos < 2e-6 and dvel < 2e-4
    l1 = np.load(os.path.join(out, "log_w1_r0.npy"))
    lw = [np.load(os.path.join(out, "log_w%d_r%d.npy" % (world, r))) for r in range(world)]
    for l in lw:
        np.testing.assert_array_equal(l1[:, :3], l[:, :3])
    assert (sum(l[:, 4] for l in lw) == n ** 3).all()


def test_single_rank_slab_equals_plain_oracle_step(tmp_path):
    """world = 1: the phase-by-phase choreography is the reference main-loop body."""
    import slab_util
    from oracle import Oracle
    from wcsph_amd.slab import SlabSim
    liquid, solid, bounds = slab_util.test_scene(7)
    sim = SlabSim(lambda **kw: slab_util.OracleBackend(**kw), liquid, solid, bounds, [0, 10 ** 6], len(liquid), 0, 1)
    o = Oracle(np.concatenate([liquid, solid]), len(liquid), 0.025, mode="clean", maxNeighbour=640,
               bounds=(bounds[1], bounds[0]))
    o.reset_param()
    for _ in range(3):
        sim.step(); o.dfsph_step()
        assert (sim.vs_iter, sim.dv_iter, sim.pr_iter) == (o.vs_iter, o.dv_iter, o.pr_iter)
        assert float(sim.dt) == pytest.approx(o.deltaT, rel=1e-6)
        r = sim.rows.numpy()
        assert np.abs(r[:, :3] - o.pos[:len(liquid)]).max() < 1e-6
        assert np.abs(r[:, 3:6] - o.vel).max() < 1e-4
    sim.B.close()


def test_balanced_cuts():
    from wcsph_amd.slab import balanced_cuts
    hist = np.array([0, 0, 5, 5, 5, 5, 5, 5, 5, 5, 0, 0])
    assert balanced_cuts(hist, 2) == [0, 6, 12]
    assert balanced_cuts(hist, 4) == [0, 4, 6, 8, 12]
    c = balanced_cuts(np.array([100, 0, 0, 0, 0, 0]), 3)
    assert c[0] == 0 and c[-1] == 6 and all(b - a >= 2 for a, b in zip(c[:-1], c[1:]))
    c = balanced_cuts(np.r_[np.zeros(30), 50 * np.ones(4), np.zeros(30)].astype(int), 4, 6)    # SlabSim's minimum slab width
    assert c[0] == 0 and c[-1] == 64 and all(b - a >= 6 for a, b in zip(c[:-1], c[1:]))
