"""The 16-bit delta neighbour streams (csrc/device.h "neighbour streams"): decoded, they are exactly the in-support part
of the reference's list -- also where the jump to the next outer tile plane does not fit 16 bits and hop slots bridge it."""
import numpy as np
import pytest

from oracle import Oracle
from util import GpuSim, make_pair, rel_err

pytestmark = pytest.mark.gpu
R = 0.025
H = 4 * R


def in_support_lists(o, i):
    """oracle list of particle i restricted to r <= h (what every kernel-weighted loop of the reference sees)"""
    ref = o.neighbor[i, :o.neighborCount[i]]
    d = o.pos[ref] - o.pos[i]
    keep = np.einsum("ij,ij->i", d, d) <= np.float32(H * H * 1.00001)
    return np.sort(ref[keep])


@pytest.mark.parametrize("mode", ["faithful", "clean"])
def test_streams_decode_to_in_support_list(mode):
    g, o = make_pair(8, mode, jitter=0.4)
    g.pd.hash_grid.update_grid(); o.update_grid()
    n = g.pd.liquid_count
    counts, lists = g.pd.hash_grid.neighbor.compact_rows(0, n)
    borderline = 0
    for i in range(0, n, 5):
        mine = np.sort(lists[i][lists[i] >= 0])
        ref = in_support_lists(o, i)
        if len(mine) != len(ref) or not np.array_equal(mine, ref):
            # an entry exactly at the list radius may fall on either side of the float comparison
            sd = set(mine.tolist()) ^ set(ref.tolist())
            assert all(abs(np.linalg.norm(o.pos[j] - o.pos[i]) - H) < 1e-5 for j in sd), (i, sd)
            borderline += 1
        assert counts[i, 0] + counts[i, 1] == len(mine)
    assert borderline <= n // 50
    g.close()


def slab_scene(nx=20, ny=20, nz=1000, wall_layers=2):
    """nx x ny x nz liquid lattice (pitch 2R) against a wall: cells are sorted in tiles of 2 x 4 x 16 with the longer of
    x / y as the outer axis, so one outer tile plane holds 2 * 20 * 1000 = 40 000 particles and the step from a
    particle to its neighbours in the next plane exceeds the 32 767 a 16-bit delta can hold."""
    ax = lambda n: (np.arange(n, dtype=np.float32) + 0.5) * np.float32(2 * R)   # noqa: E731
    X, Y, Z = np.meshgrid(ax(nx), ax(ny), ax(nz), indexing="ij")
    liquid = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1).astype(np.float32)
    xw = -(np.arange(wall_layers, dtype=np.float32) + 0.5) * np.float32(2 * R)
    X, Y, Z = np.meshgrid(xw, ax(ny), ax(nz), indexing="ij")
    solid = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1).astype(np.float32)
    return liquid, solid


@pytest.mark.parametrize("mode", ["clean", "faithful"])
def test_hop_slots_on_wide_planes(mode):
    liquid, solid = slab_scene()
    g = GpuSim(liquid, solid, mode=mode)
    o = Oracle(g.pd.pos.to_numpy(), len(liquid), R, mode=mode)
    g.pd.hash_grid.update_grid(); o.update_grid()
    n = len(liquid)
    np.testing.assert_array_equal(g.pd.hash_grid.neighborCount.to_numpy(), o.neighborCount)
    probe = np.r_[0:64, n // 2:n // 2 + 64, n - 64:n]
    hops = 0
    for i in probe:
        c, l = g.pd.hash_grid.neighbor.compact_rows(int(i), 1)
        mine = np.sort(l[0][l[0] >= 0])
        ref = in_support_lists(o, int(i))
        sd = set(mine.tolist()) ^ set(ref.tolist())
        assert all(abs(np.linalg.norm(o.pos[j] - o.pos[i]) - H) < 1e-5 for j in sd), (i, sd)
        hops += c[0, 2]
    assert hops > 0                     # the scene must actually exercise the hop slots
    # and the passes that iterate the streams agree with the oracle
    g.call("dfsph_compute_density"); o.compute_density()
    g.call("dfsph_compute_coff"); o.compute_dfsph_coff()
    assert np.abs(g.pd.rho.to_numpy() - o.rho).max() / o.rho.max() < 1e-5
    assert rel_err(g.alpha_coff.to_numpy(), o.alpha_coff) < 2e-4
    g.call("dfsph_reset_param"); o.reset_param()
    g.call("dfsph_step"); o.dfsph_step()
    assert g.iters() == (o.vs_iter, o.dv_iter, o.pr_iter)
    assert np.abs(g.pd.pos.to_numpy()[:n] - o.pos[:n]).max() < 2e-6
    assert rel_err(g.pd.vel.to_numpy(), o.vel) < 2e-3
    g.close()
