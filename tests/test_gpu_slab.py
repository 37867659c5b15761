"""GPU: the slab path with the real HIP backend.  (a) one rank, phase-by-phase, against the checker backend;
(b) two gloo ranks SHARING the one GPU of the test box (halo buffers staged through the host) against one rank."""
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
STEPS, VX = 6, 4.0


def _hip_run(rank, world, port, outdir, n):
    import slab_util
    from wcsph_amd.slab import SlabSim, HipBackend
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    liquid, solid, bounds = slab_util.test_scene(n)
    cuts, parts = slab_util.split_scene(liquid, bounds, world)
    sim = SlabSim(lambda **kw: HipBackend(**kw), parts[rank], solid, bounds, cuts, len(liquid), rank, world,
                  device=dev, comm_device=torch.device("cpu"), gid_local=slab_util.split_scene.gids[rank],
                  neighbor_mode=os.environ.get("SLAB_TEST_MODE", "clean"))
    sim.rows[:, 3] = VX
    torch.cuda.synchronize()
    log = []
    for k in range(STEPS):
        if world > 1 and k == int(os.environ.get("SLAB_TEST_REBALANCE_AT", "-1")):
            assert sim.rebalance(force=True)
        sim.step()
        log.append((sim.vs_iter, sim.dv_iter, sim.pr_iter, float(sim.dt), sim.n_owned))
    torch.cuda.synchronize()
    np.save(os.path.join(outdir, "hrows_w%d_r%d.npy" % (world, rank)), sim.rows.cpu().numpy())
    np.save(os.path.join(outdir, "hlog_w%d_r%d.npy" % (world, rank)), np.array(log))
    np.save(os.path.join(outdir, "hwaits_w%d_r%d.npy" % (world, rank)), np.array([sim.n_host_waits, sim.n_exchanges, STEPS, getattr(sim, "n_resident", 0)]))
    sim.B.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_hip_slab_one_rank_matches_checker(tmp_path):
    import slab_util
    from wcsph_amd.slab import SlabSim
    n = 8
    _hip_run(0, 1, 0, str(tmp_path), n)
    liquid, solid, bounds = slab_util.test_scene(n)
    ref = SlabSim(lambda **kw: slab_util.OracleBackend(**kw), liquid, solid, bounds, [0, 10 ** 6], len(liquid), 0, 1)
    ref.rows[:, 3] = VX
    rlog = []
    for _ in range(STEPS):
        ref.step()
        rlog.append((ref.vs_iter, ref.dv_iter, ref.pr_iter, float(ref.dt), ref.n_owned))
    got = np.load(os.path.join(str(tmp_path), "hrows_w1_r0.npy"))
    exp = ref.rows.numpy()
    assert np.abs(got[:, :3] - exp[:, :3]).max() < 5e-6
    assert np.abs(got[:, 3:6] - exp[:, 3:6]).max() < 1e-3
    glog = np.load(os.path.join(str(tmp_path), "hlog_w1_r0.npy"))
    np.testing.assert_array_equal(glog[:, :3], np.array(rlog)[:, :3])
    np.testing.assert_allclose(glog[:, 3], np.array(rlog)[:, 3], rtol=1e-6)


@pytest.mark.parametrize("world,n", [(2, 8), (3, 12)])
def test_hip_slab_ranks_match_one_rank(tmp_path, world, n):
    """world = 3: an INTERIOR rank (two neighbours, ghosts and migration on both sides); the block drifts at 4 m/s, so
    rows cross both cuts during the run"""
    import slab_util
    from test_slab_gloo import spawn_ranks
    out = str(tmp_path)
    _hip_run(0, 1, 0, out, n)
    spawn_ranks(_hip_run, world, out, n)
    one = np.load(os.path.join(out, "hrows_w1_r0.npy"))
    many = np.concatenate([np.load(os.path.join(out, "hrows_w%d_r%d.npy" % (world, r))) for r in range(world)])
    assert len(one) == len(many) == n ** 3
    dpos, dvel, _ = slab_util.match_rows(one, many)
    assert dpos < 5e-6 and dvel < 1e-3
    l1 = np.load(os.path.join(out, "hlog_w1_r0.npy"))
    lw = [np.load(os.path.join(out, "hlog_w%d_r%d.npy" % (world, r))) for r in range(world)]
    for l in lw:
        np.testing.assert_array_equal(l1[:, :3], l[:, :3])
    assert (sum(l[:, 4] for l in lw) == n ** 3).all()
    assert sum(l[0, 4] != l[-1, 4] for l in lw) >= 2      # rows crossed a cut: at least two ranks gained or lost some


def test_faithful_slabs_reproduce_the_single_context(tmp_path):
    """Faithful neighbour semantics across ranks: the reference's hash table is global (`count` buckets, HashGrid.py:114),
    so every rank counts its owned liquid particles per bucket and the histogram is all-reduced each step (two 16-bit
    counts per word), the static boundary histogram once; the CFL prefix (Q5) runs on global particle ids.  Two ranks
    (with a forced rebalance on the way) must reproduce one rank, and one rank must reproduce the plain library step --
    density to the north-star 1e-4."""
    import slab_util
    from test_slab_gloo import spawn_ranks
    from util import GpuSim
    n, out = 10, str(tmp_path)
    os.environ.update(SLAB_TEST_MODE="faithful", SLAB_TEST_REBALANCE_AT="3")
    try:
        _hip_run(0, 1, 0, out, n)
        spawn_ranks(_hip_run, 2, out, n)
    finally:
        os.environ.pop("SLAB_TEST_MODE", None); os.environ.pop("SLAB_TEST_REBALANCE_AT", None)
    one = np.load(os.path.join(out, "hrows_w1_r0.npy"))
    two = np.concatenate([np.load(os.path.join(out, "hrows_w2_r%d.npy" % r)) for r in range(2)])
    dpos, dvel, _ = slab_util.match_rows(one, two)
    assert dpos < 5e-6 and dvel < 1e-3
    l1 = np.load(os.path.join(out, "hlog_w1_r0.npy"))
    l2 = np.load(os.path.join(out, "hlog_w2_r0.npy"))
    np.testing.assert_array_equal(l1[:, :3], l2[:, :3])
    np.testing.assert_allclose(l1[:, 3], l2[:, 3], rtol=1e-6)            # same dt history: the CFL prefix saw the same particles
    # the plain library step on the same scene
    liquid, solid, bounds = slab_util.test_scene(n)
    g = GpuSim(liquid, solid, mode="faithful")
    g.call("dfsph_reset_param")
    v = np.zeros((len(liquid), 3), np.float32); v[:, 0] = VX
    g.pd.vel.from_numpy(v)
    for _ in range(STEPS):
        g.call("dfsph_step")
    pos = g.pd.pos.to_numpy()[:len(liquid)]
    order = np.argsort(one[:, 14].copy().view(np.int32))
    assert np.abs(one[order, :3] - pos).max() < 5e-6
    assert np.abs(one[order, 3:6] - g.pd.vel.to_numpy()).max() < 1e-3
    assert g.iters() == tuple(int(x) for x in l1[-1, :3])
    g.close()


def test_overlapped_halo_exchange_changes_nothing(tmp_path):
    """WCSPH_SLAB_OVERLAP: every neighbour pass that needs a halo refresh runs on the slab interior while the exchange is
    in flight and on the rest afterwards -- same particles, same inputs, same results as exchange-then-pass."""
    from test_slab_gloo import spawn_ranks
    n = 14
    outs = []
    for flag in ("0", "1"):
        out = os.path.join(str(tmp_path), "ov" + flag)
        os.makedirs(out)
        os.environ["WCSPH_SLAB_OVERLAP"] = flag
        try:
            spawn_ranks(_hip_run, 2, out, n)
        finally:
            os.environ.pop("WCSPH_SLAB_OVERLAP", None)
        outs.append(out)
    rows = [np.concatenate([np.load(os.path.join(o, "hrows_w2_r%d.npy" % r)) for r in range(2)]) for o in outs]
    logs = [np.load(os.path.join(o, "hlog_w2_r0.npy")) for o in outs]
    np.testing.assert_array_equal(logs[0][:, :3], logs[1][:, :3])
    assert len(rows[0]) == len(rows[1]) == n ** 3
    import slab_util
    dpos, dvel, _ = slab_util.match_rows(rows[0], rows[1])
    assert dpos < 1e-6 and dvel < 1e-4
    # the split is not vacuous on this scene: each rank's slab is wider than its two boundary strips
    liquid, solid, bounds = slab_util.test_scene(n)
    cuts, _ = slab_util.split_scene(liquid, bounds, 2)
    assert cuts[1] - cuts[0] > 4 and cuts[2] - cuts[1] > 4


@pytest.mark.parametrize("mode", ["clean", "faithful"])
def test_native_slab_step_equals_the_python_phase_driver(tmp_path, mode):
    """wcsph_slab_step (the whole choreography inside the library, the communicator reached through callbacks) against
    SlabSim._step's phase-by-phase Python form (WCSPH_SLAB_NATIVE=0), two ranks: same launches in the same order, so the
    same iteration counts, dt history and -- up to the order of the atomics -- the same rows."""
    import slab_util
    from test_slab_gloo import spawn_ranks
    n, outs = 10, []
    os.environ["SLAB_TEST_MODE"] = mode
    try:
        for native in ("1", "0"):
            out = os.path.join(str(tmp_path), "native" + native)
            os.makedirs(out)
            os.environ["WCSPH_SLAB_NATIVE"] = native
            spawn_ranks(_hip_run, 2, out, n)
            outs.append(out)
    finally:
        os.environ.pop("WCSPH_SLAB_NATIVE", None); os.environ.pop("SLAB_TEST_MODE", None)
    rows = [np.concatenate([np.load(os.path.join(o, "hrows_w2_r%d.npy" % r)) for r in range(2)]) for o in outs]
    logs = [np.load(os.path.join(o, "hlog_w2_r0.npy")) for o in outs]
    np.testing.assert_array_equal(logs[0][:, :3], logs[1][:, :3])
    np.testing.assert_allclose(logs[0][:, 3], logs[1][:, 3], rtol=1e-7)
    dpos, dvel, _ = slab_util.match_rows(rows[0], rows[1])
    assert dpos < 1e-6 and dvel < 1e-4


@pytest.mark.parametrize("mode", ["clean", "faithful"])
def test_speculative_slab_step_changes_nothing_and_hides_the_decisions(tmp_path, mode):
    """wcsph_slab_step queues the common-case continuation behind every solver decision (the decisions are taken on the device
    from the all-reduced, rank-identical sums; gated launches, unconditional exchanges): against WCSPH_SLAB_SPECULATE=0 -- one
    host wait per decision, the control flow of rounds 1-3 -- the same iteration counts, dt history and rows; and the host
    waits with an idle GPU behind them drop from >= 4 per step to the step's last mail (+ 1 for the row classification)
    wherever the common case holds."""
    import slab_util
    from test_slab_gloo import spawn_ranks
    n, outs = 12, []
    os.environ["SLAB_TEST_MODE"] = mode
    try:
        for spec in ("1", "0"):
            out = os.path.join(str(tmp_path), "spec" + spec)
            os.makedirs(out)
            os.environ["WCSPH_SLAB_SPECULATE"] = spec
            spawn_ranks(_hip_run, 2, out, n)
            outs.append(out)
    finally:
        os.environ.pop("WCSPH_SLAB_SPECULATE", None); os.environ.pop("SLAB_TEST_MODE", None)
    rows = [np.concatenate([np.load(os.path.join(o, "hrows_w2_r%d.npy" % r)) for r in range(2)]) for o in outs]
    logs = [np.load(os.path.join(o, "hlog_w2_r0.npy")) for o in outs]
    np.testing.assert_array_equal(logs[0][:, :3], logs[1][:, :3])
    np.testing.assert_allclose(logs[0][:, 3], logs[1][:, 3], rtol=1e-7)
    dpos, dvel, _ = slab_util.match_rows(rows[0], rows[1])
    assert dpos < 1e-6 and dvel < 1e-4
    w_spec, w_plain = (np.load(os.path.join(o, "hwaits_w2_r0.npy")) for o in outs)
    steps = float(w_spec[2])
    # plain: divergence, viscosity, CFL, pressure (>= 4) + the classification; speculative: the last mail + the classification,
    # plus one per decision that fell the other way (the drifting block of this scene needs extra iterations in some steps)
    assert w_plain[0] / steps >= 5.0
    assert w_spec[0] / steps <= w_plain[0] / steps - 2.0
    print("host waits per step: speculative %.2f, plain %.2f; exchanges per step %.1f / %.1f" % (w_spec[0] / steps, w_plain[0] / steps, w_spec[1] / steps, w_plain[1] / steps))


def test_tail_classification_saves_a_host_wait_and_changes_nothing(tmp_path):
    """(round 6) The classification of the next step's migration / ghost rebuild is taken in the TAIL of wcsph_slab_step, behind
    the step's last gate, and its counts travel with one more mail queued right behind the last one: where a step ends on its
    speculative path the driver finds them on the host (wcsph_slab_tail_classify_result) instead of waiting for a launch of its
    own.  Three ranks, rows crossing both cuts: against WCSPH_SLAB_TAIL_CLASSIFY=0 the same rows, iteration counts and dt, and
    fewer host waits."""
    import slab_util
    from test_slab_gloo import spawn_ranks
    n, outs = 12, []
    try:
        for tail in ("1", "0"):
            out = os.path.join(str(tmp_path), "tail" + tail)
            os.makedirs(out)
            os.environ["WCSPH_SLAB_TAIL_CLASSIFY"] = tail
            spawn_ranks(_hip_run, 3, out, n)
            outs.append(out)
    finally:
        os.environ.pop("WCSPH_SLAB_TAIL_CLASSIFY", None)
    rows = [np.concatenate([np.load(os.path.join(o, "hrows_w3_r%d.npy" % r)) for r in range(3)]) for o in outs]
    logs = [np.load(os.path.join(o, "hlog_w3_r1.npy")) for o in outs]
    np.testing.assert_array_equal(logs[0][:, :3], logs[1][:, :3])
    np.testing.assert_allclose(logs[0][:, 3], logs[1][:, 3], rtol=1e-7)
    own = [np.load(os.path.join(outs[0], "hlog_w3_r%d.npy" % r))[:, 4] for r in range(3)]
    assert sum(o[0] != o[-1] for o in own) >= 2               # rows migrated: at least two ranks gained or lost some
    dpos, dvel, _ = slab_util.match_rows(rows[0], rows[1])
    assert dpos < 1e-6 and dvel < 1e-4
    w_on, w_off = (np.load(os.path.join(o, "hwaits_w3_r1.npy")) for o in outs)
    steps = float(w_on[2])
    print("host waits per step (rank 1 of 3): tail classification %.2f, without %.2f" % (w_on[0] / steps, w_off[0] / steps))
    assert w_on[0] / steps <= w_off[0] / steps - 0.5          # (every step that ends on its speculative path saves one)


def test_native_classifier_equals_the_torch_classification():
    """wcsph_slab_classify (one launch) against the torch form of SlabSim.redistribute: the same four row sets and the same
    five counts, cell columns formed as the grid update forms them (ref: HashGrid.py:68), including rows exactly on a cut"""
    import slab_util
    from wcsph_amd.slab import HipBackend, GHOST_CELLS, STATE_W
    dev = torch.device("cuda", 0)
    liquid, solid, bounds = slab_util.test_scene(8)
    B = HipBackend(capacity=20000, solid_local=solid, bounds=bounds, window=(0, 17), particleR=0.025, device=dev)
    minx, inv = np.float32(bounds[0][0]), np.float32(1.0 / (2 * 0.025))       # HashGrid.py:46: gridR = 2 R
    g = torch.Generator(device="cpu").manual_seed(5)
    n = 150001
    rows = torch.rand((n, STATE_W), generator=g)
    lo, hi = 7, 19
    rows[:, 0] = float(minx) + rows[:, 0] * float(26.0 / inv)
    rows[:64, 0] = float(minx) + torch.arange(64, dtype=torch.float32) * float(np.float32(0.5) / inv)     # cell edges
    rows = rows.to(dev)
    cx = torch.trunc((rows[:, 0] - float(minx)) * float(inv)).to(torch.int64)
    torch.cuda.synchronize()
    for has_l, has_r in ((1, 1), (0, 1), (1, 0), (0, 0)):
        with torch.cuda.stream(B.stream):       # as SlabSim.step does: the library's launches and torch's share one stream
            got = B.classify_rows(rows, lo, hi, GHOST_CELLS, has_l, has_r)
            torch.cuda.synchronize()
        assert got is not None
        cnt, lists = got
        G = GHOST_CELLS
        f = torch.zeros_like(cx, dtype=torch.bool)
        go_l = (cx < lo) & (cx >= lo - G) if has_l else f
        go_r = (cx >= hi) & (cx < hi + G) if has_r else f
        lost = ((cx < lo - G) if has_l else f) | ((cx >= hi + G) if has_r else f)
        near_l = (cx >= lo) & (cx < lo + G) & ~go_r & ~lost if has_l else f
        near_r = (cx < hi) & (cx >= hi - G) & ~go_l & ~lost if has_r else f
        for want, have in zip((near_l, near_r, go_l, go_r), lists):
            assert torch.equal(torch.nonzero(want).flatten(), torch.sort(have)[0])
        assert cnt == [int(near_l.sum()), int(near_r.sum()), int(go_l.sum()), int(go_r.sum()), int(lost.sum())]
    B.close()


def test_resident_slab_state_equals_the_row_table_round_trip(tmp_path):
    """Round 5: between two steps the state stays in the library -- the driver classifies the context's own rows, takes out the
    ~6 % that travel and applies what arrives (wcsph_state_classify / _gather_rows / _apply; dead slots vanish in the next sort)
    -- instead of storing and reloading every row (WCSPH_SLAB_RESIDENT=0: rounds 2-4).  Three ranks, rows crossing both cuts of the
    interior rank, a forced rebalance on the way: the two forms agree to rounding level, and the resident form really ran."""
    import slab_util
    from test_slab_gloo import spawn_ranks
    n, world = 12, 3
    res = {}
    for flag in ("1", "0"):
        out = os.path.join(str(tmp_path), "resident" + flag)
        os.makedirs(out)
        os.environ.update(WCSPH_SLAB_RESIDENT=flag, SLAB_TEST_REBALANCE_AT="3")
        try:
            spawn_ranks(_hip_run, world, out, n)
        finally:
            os.environ.pop("WCSPH_SLAB_RESIDENT", None); os.environ.pop("SLAB_TEST_REBALANCE_AT", None)
        rows = np.concatenate([np.load(os.path.join(out, "hrows_w%d_r%d.npy" % (world, r))) for r in range(world)])
        logs = [np.load(os.path.join(out, "hlog_w%d_r%d.npy" % (world, r))) for r in range(world)]
        used = [int(np.load(os.path.join(out, "hwaits_w%d_r%d.npy" % (world, r)))[3]) for r in range(world)]
        res[flag] = (rows, logs, used)
    assert min(res["1"][2]) >= STEPS - 2 and max(res["0"][2]) == 0          # every step but the first and the one after the rebalance
    dpos, dvel, _ = slab_util.match_rows(res["1"][0], res["0"][0])
    assert dpos < 1e-6 and dvel < 1e-4
    for la, lb in zip(res["1"][1], res["0"][1]):
        np.testing.assert_array_equal(la[:, :3], lb[:, :3])                   # iteration counts
        np.testing.assert_array_equal(la[:, 4], lb[:, 4])                     # owned rows per step: the same migrations
