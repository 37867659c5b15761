"""Edge cases of the boundary (empty / ragged / degenerate inputs) on the GPU, against the oracle where it applies."""
import numpy as np
import pytest

from util import GpuSim, rel_err

pytestmark = pytest.mark.gpu
BOX = np.array([[-0.5, 0.0, -0.5], [0.5, 1.0, 0.5]])   # two far solids that only fix the bbox


def _step_both(liquid, solid, mode, steps=2, **kw):
    from oracle import Oracle
    g = GpuSim(liquid, solid, mode=mode, **kw)
    o = Oracle(g.pd.pos.to_numpy(), len(liquid), 0.025, mode=mode)
    g.call("dfsph_reset_param"); o.reset_param()
    for _ in range(steps):
        g.call("dfsph_step"); o.dfsph_step()
    return g, o


@pytest.mark.parametrize("mode", ["faithful", "clean"])
def test_single_liquid_particle_free_fall(mode):
    g, o = _step_both(np.array([[0.0, 0.5, 0.0]]), BOX, mode, steps=3)
    assert g.pd.hash_grid.neighborCount.to_numpy()[0] == o.neighborCount[0]
    np.testing.assert_allclose(g.pd.pos.to_numpy()[0], o.pos[0], atol=1e-7)
    np.testing.assert_allclose(g.pd.vel.to_numpy()[0], o.vel[0], rtol=1e-6, atol=1e-9)
    assert g.pd.vel.to_numpy()[0, 1] < 0.0          # it falls
    assert g.iters() == (o.vs_iter, o.dv_iter, o.pr_iter)
    g.close()


def test_no_liquid_at_all():
    g = GpuSim(np.zeros((0, 3)), BOX, mode="faithful")
    g.call("dfsph_reset_param")
    g.call("grid_update")
    g.call("dfsph_compute_density")
    assert g.pd.rho.to_numpy().shape == (0,)
    assert g.pd.pos.to_numpy().shape == (2, 3)
    g.pd.sync()
    g.close()


@pytest.mark.parametrize("n", [63, 65, 257, 1000])
def test_ragged_counts(n):
    """particle counts that are not multiples of the wave (64) / block (256) size"""
    rng = np.random.default_rng(n)
    liquid = rng.uniform([-0.2, 0.1, -0.2], [0.2, 0.5, 0.2], (n, 3))
    # thin the random cloud: no two particles closer than 0.6 R (a reference run would explode the same way otherwise)
    keep = [0]
    for i in range(1, n):
        if np.min(np.linalg.norm(liquid[keep] - liquid[i], axis=1)) > 0.015:
            keep.append(i)
    liquid = liquid[keep]
    g, o = _step_both(liquid, BOX, "clean", steps=1)
    np.testing.assert_array_equal(g.pd.hash_grid.neighborCount.to_numpy(), o.neighborCount)
    assert np.abs(g.pd.rho.to_numpy() - o.rho).max() / o.rho.max() < 1e-5
    assert np.abs(g.pd.pos.to_numpy() - o.pos).max() < 1e-6
    g.close()


def test_no_solids_in_range_and_flat_grid():
    """a one-cell-thick grid in z (all particles share one z cell) and no boundary particle near the liquid"""
    gx = np.arange(12) * 0.05
    liquid = np.stack(np.meshgrid(gx, gx + 0.2, [0.0], indexing="ij"), -1).reshape(-1, 3)
    solid = np.array([[-1.0, 0.0, 0.0], [2.0, 2.0, 0.0]])      # same z: blockSize z = 1
    g, o = _step_both(liquid, solid, "faithful", steps=2)
    assert g.pd.hash_grid.blockSize.to_numpy()[0, 2] == 1 == o.blockSize[2]
    np.testing.assert_array_equal(g.pd.hash_grid.neighborCount.to_numpy(), o.neighborCount)
    assert np.abs(g.pd.rho.to_numpy() - o.rho).max() / o.rho.max() < 1e-5
    assert np.abs(g.pd.pos.to_numpy() - o.pos).max() < 1e-6
    g.close()


def test_all_liquid_outside_the_box():
    """Q4: the grid box is fixed at setup; particles moved outside afterwards get no neighbours and free-fall"""
    liquid = np.array([[0.0, 0.5, 0.0], [0.02, 0.5, 0.0], [0.0, 0.52, 0.0]])
    from oracle import Oracle
    g = GpuSim(liquid, BOX, mode="faithful")
    o = Oracle(g.pd.pos.to_numpy(), 3, 0.025, mode="faithful")
    p = g.pd.pos.to_numpy(); p[:3] += 7.0
    g.pd.pos.from_numpy(p); o.pos[:3] = p[:3]
    g.call("dfsph_reset_param"); o.reset_param()
    g.call("dfsph_step"); o.dfsph_step()
    assert (g.pd.hash_grid.neighborCount.to_numpy() == 0).all() and (o.neighborCount == 0).all()
    np.testing.assert_allclose(g.pd.pos.to_numpy()[:3], o.pos[:3], atol=1e-6)
    g.close()


def test_call_order_errors():
    from wcsph_amd._lib import WcsphError
    g = GpuSim(np.array([[0.0, 0.5, 0.0]]), BOX, mode="clean")
    with pytest.raises(WcsphError) as e:
        g.call("dfsph_compute_density")          # before any update_grid()
    assert e.value.code == -1
    g.pd.hash_grid.update_grid()
    g.call("dfsph_compute_density")
    p = g.pd.pos.to_numpy()
    g.pd.pos.from_numpy(p)                       # positions rewritten: cells are stale again
    with pytest.raises(WcsphError):
        g.call("dfsph_compute_density")
    with pytest.raises(ValueError):
        g.pd.vel.from_numpy(np.zeros((5, 3), np.float32))   # wrong shape
    g.close()


def test_compact_list_capacity_is_an_error_not_silent():
    """a pathological clump: more in-support neighbours than the compact list holds"""
    from wcsph_amd._lib import WcsphError
    rng = np.random.default_rng(0)
    liquid = np.array([0.0, 0.5, 0.0]) + rng.uniform(-0.02, 0.02, (400, 3))
    g = GpuSim(liquid, BOX, mode="clean", maxInGrid=512)
    g.pd.hash_grid.update_grid()
    with pytest.raises(WcsphError) as e:
        g.pd.sync()
    assert e.value.code == -4
    g.close()


def test_results_are_reproducible_to_rounding():
    """Two runs differ at most by fp32 summation order (atomic slot order inside a cell, Q16)."""
    from util import small_scene
    liquid, solid = small_scene(8, jitter=0.3)
    out = []
    for _ in range(2):
        g = GpuSim(liquid, solid, mode="faithful")
        g.call("dfsph_reset_param")
        for _ in range(3):
            g.call("dfsph_step")
        out.append((g.pd.pos.to_numpy(), g.pd.rho.to_numpy(), g.iters()))
        g.close()
    assert out[0][2] == out[1][2]
    assert np.abs(out[0][0] - out[1][0]).max() < 1e-6
    assert rel_err(out[0][1], out[1][1]) < 1e-5


@pytest.mark.parametrize("mode", ["faithful", "clean"])
def test_large_sparse_grid_takes_the_prefixed_scan(mode):
    """beyond ~16 M cells the cell-counter scan prefix-sums its block totals once (grid.h k_scan_bsums, SCAN_BPRE_MIN)
    instead of having every block re-read them.  Two small liquid blocks in opposite corners of a 300^3-cell box (27 M cells, 13 k
    scan blocks, the occupied bounding box spans the domain so that every block is active): raw list lengths, bucket occupancy and one
    DFSPH step against the oracle (ref: HashGrid.py:56-106)."""
    from oracle import Oracle
    R, L = 0.025, 14.9
    big = np.array([[0.0, 0.0, 0.0], [L, L, L]])
    k = np.arange(6 ** 3)
    blk = np.stack([k // 36, (k // 6) % 6, k % 6], 1) * (2 * R) + 0.31
    rng = np.random.default_rng(5)
    liquid = np.concatenate([blk, L - 0.3 - blk]) + rng.uniform(-0.2, 0.2, (2 * len(blk), 3)) * R
    g = GpuSim(liquid, big, mode=mode)
    assert int(np.prod(g.pd.hash_grid.blockSize.to_numpy()[0].astype(np.int64))) > 8192 * 2048           # more scan blocks than SCAN_BPRE_MIN
    o = Oracle(g.pd.pos.to_numpy(), len(liquid), R, mode=mode)
    g.pd.hash_grid.update_grid(); o.update_grid()
    np.testing.assert_array_equal(g.pd.hash_grid.neighborCount.to_numpy(), o.neighborCount)
    if mode == "faithful":
        np.testing.assert_array_equal(g.pd.hash_grid.gridCount, o.gridCount)
    g.call("dfsph_reset_param"); o.reset_param()
    g.call("dfsph_step"); o.dfsph_step()
    assert rel_err(g.pd.rho.to_numpy(), o.rho) < 1e-5
    assert np.abs(g.pd.pos.to_numpy()[:len(liquid)] - o.pos[:len(liquid)]).max() < 2e-6
    g.close(); o.close()
