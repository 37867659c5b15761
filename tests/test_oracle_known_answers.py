"""Pins the CPU oracle (oracle/wcsph_ref.c) with the analytic known-answers of SURVEY Appendix C.

The reference ships no tests or golden vectors and cannot run offline (no taichi), so these are the
only reference-independent anchors: kernel normalisation, closed-form lattice sums, int32-wrap hash
values computed by hand from the reference formula (HashGrid.py:108-114), scene set-up facts."""
import math

import numpy as np
import pytest

from oracle import Oracle
from wcsph_amd import scene


def _dummy(count, n_liquid=4, mode="faithful", particleR=0.025):
    pos = np.zeros((count, 3), np.float32)
    pos[:, 0] = np.linspace(0, 1, count)
    return Oracle(pos, n_liquid, particleR, mode=mode, maxNeighbour=8)


def test_kernel_normalisation_and_samples():
    o = _dummy(16)
    h = 0.1
    assert o.W_norm(0.0) == pytest.approx(2546.479089, rel=1e-6)           # 8/(pi h^3)
    assert o.W_norm(0.5 * h) == pytest.approx(0.25 * o.W_norm(0.0), rel=1e-6)
    assert o.W_norm(h) == 0.0 and o.W_norm(1.01 * h) == 0.0
    r = np.linspace(0, h, 20001)
    w = np.array([o.W_norm(x) for x in r], np.float64)
    f = 4 * math.pi * r * r * w
    integral = float(np.sum(0.5 * (f[1:] + f[:-1]) * np.diff(r)))
    assert integral == pytest.approx(1.0, abs=2e-5)                        # int W dV = 1
    np.testing.assert_allclose(o.gradW([0.05, 0, 0]), [-38197.186, 0, 0], rtol=1e-6, atol=1e-3)
    np.testing.assert_allclose(o.gradW([0.03, 0.04, 0]), [-22918.312, -30557.749, 0], rtol=2e-6, atol=1e-3)
    # gradW continuous at q = 0.5: both branches give -0.25 m_l h3 / h
    lo, hi = o.gradW([0.05 * (1 - 1e-6), 0, 0])[0], o.gradW([0.05 * (1 + 1e-6), 0, 0])[0]
    assert lo == pytest.approx(hi, rel=1e-4)
    assert np.all(o.gradW([0, 0, 0]) == 0) and np.all(o.gradW([0.2, 0, 0]) == 0)
    assert o.VL0 == pytest.approx(1.0e-4, rel=1e-6) and o.liqiudMass == pytest.approx(0.1, rel=1e-6)
    assert (o.gridR, o.searchR) == (pytest.approx(0.05), pytest.approx(0.1))


def test_cohesion_adhesion_constants():
    o = _dummy(16)
    h = 0.1
    m_k, m_c = 32.0 / (math.pi * h ** 9), h ** 6 / 64.0
    assert m_k == pytest.approx(1.0185916e10, rel=1e-6) and m_c == pytest.approx(1.5625e-8, rel=1e-6)
    r = 0.07
    assert o.coh_W(r) == pytest.approx(m_k * (h - r) ** 3 * r ** 3, rel=1e-4)
    r = 0.03   # inner branch: "- m_c" sits OUTSIDE the m_k factor (Q9)
    assert o.coh_W(r) == pytest.approx(m_k * 2 * (h - r) ** 3 * r ** 3 - m_c, rel=1e-4)
    assert 0.007 / h ** 3.25 == pytest.approx(12.4479559, rel=1e-6)
    r = 0.08
    assert o.adh_W(r) == pytest.approx(12.4479559 * (-4 * r * r / h + 6 * r - 2 * h) ** 0.25, rel=1e-4)
    assert o.adh_W(0.04) == 0.0 and o.adh_W(0.11) == 0.0


@pytest.mark.parametrize("count,expect", [
    (33387, {(1, 2, 3): 27209, (20, 4, 20): 5829, (40, 40, 40): 15632, (27, 0, 13): 30463}),
    (17128, {(1, 2, 3): 17070, (20, 4, 20): 6436, (40, 40, 40): 15872, (27, 0, 13): 8804}),
])
def test_hash_values_int32_wrap(count, expect):
    """(40,40,40): the products overflow int32 (-1340723576, 773986520, -955255656)."""
    o = _dummy(count)
    for cell, h in expect.items():
        assert o.cell_hash(*cell) == h
    # independent restatement with Python big ints
    def ref(x, y, z):
        w = lambda v: ((v + 2 ** 31) % 2 ** 32) - 2 ** 31   # noqa: E731
        p = w(73856093 * x) ^ w(19349663 * y) ^ w(83492791 * z)
        return ((p % count) + count) % count if p >= 0 else ((-((-p) % count)) + count) % count
    rng = np.random.default_rng(1)
    for x, y, z in rng.integers(0, 600, (200, 3)):
        assert o.cell_hash(x, y, z) == ref(int(x), int(y), int(z))


def test_infinite_lattice_density_and_alpha():
    """Interior of a cubic lattice of spacing 0.05: rho = 799.97797, alpha = -9.19463897e-3, sum gradW = 0."""
    n = 11
    g = np.arange(n) * 0.05
    P = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    for mode in ("clean",):
        o = Oracle(P, len(P), 0.025, mode=mode)
        o.update_grid(); o.compute_density(); o.compute_dfsph_coff()
        c = (n // 2) * n * n + (n // 2) * n + n // 2
        assert o.rho[c] == pytest.approx(799.97797, rel=2e-6)
        assert o.alpha_coff[c] == pytest.approx(-9.19463897e-3, rel=1e-5)
        assert o.neighborCount[c] == 124


def test_default_scene_facts():
    """dfsph.py:59-82 with n = 20: bbox (-1,0,-1)..(1,2,1), blockSize (41,41,41), bucket cap never hit."""
    liquid, solid = scene.dam_break_dfsph(20)
    assert liquid.shape == (8000, 3)
    np.testing.assert_allclose(liquid[0], [-0.475, 0.2, -0.475])
    np.testing.assert_allclose(liquid[-1], [0.475, 1.15, 0.475])
    pos = np.concatenate([liquid, solid]).astype(np.float32)
    np.testing.assert_allclose(pos.min(0), [-1, 0, -1]); np.testing.assert_allclose(pos.max(0), [1, 2, 1])
    o = Oracle(pos, 8000, 0.025, mode="faithful")
    assert o.blockSize == (41, 41, 41)
    o.update_grid()
    assert o.overflow_grid == 0 and o.overflow_neighbor == 0
    assert o.gridCount.max() <= 64
    # Q1: the raw list is several times longer than the ~27 in-support neighbours of the lattice
    assert 100 < o.neighborCount.mean() < 400
    # area density of the generated shell ~ the reference's model/box_boundry.obj (25 387 points on 24 m^2)
    assert abs(len(solid) / 24.0 - 1058.0) < 30.0
    # sesph scene counts (sesph.py:66-92)
    l2, s2 = scene.sesph_scene()
    assert (len(l2), len(s2)) == (8000, 9128)


def test_faithful_list_is_the_bucket_multiset():
    """Independent NumPy emulation of HashGrid.update_grid on a tiny jittered scene: same multiset per particle,
    and duplicates / far aliased entries really occur (otherwise 'faithful' would be untested)."""
    rng = np.random.default_rng(5)
    liquid, solid = scene.dam_break_dfsph(5)
    liquid = liquid + rng.uniform(-0.01, 0.01, liquid.shape)
    pos = np.concatenate([liquid, solid]).astype(np.float32)
    nl, n = len(liquid), len(pos)
    o = Oracle(pos, nl, 0.025, mode="faithful")
    o.update_grid()
    mn = pos.min(0)
    cells = np.trunc((pos - mn) * np.float32(1.0 / 0.05)).astype(np.int64)
    bs = np.array(o.blockSize)
    buckets = {}
    for i in range(n):
        buckets.setdefault(o.cell_hash(*cells[i]), []).append(i)
    dup_seen = far_seen = 0
    for i in range(0, nl, 3):
        exp = []
        for m in range(-2, 3):
            for q in range(-2, 3):
                for w in range(-2, 3):
                    c = cells[i] + (m, q, w)
                    if np.all(c >= 0) and np.all(c < bs):
                        exp += [j for j in buckets.get(o.cell_hash(*c), []) if j != i]
        got = o.neighbor[i, :o.neighborCount[i]]
        assert sorted(got) == sorted(exp)
        dup_seen += len(exp) != len(set(exp))
        far_seen += any(np.abs(cells[j] - cells[i]).max() > 2 for j in exp)
    assert dup_seen > 0 and far_seen > 0


def test_clean_list_is_the_cell_neighbourhood():
    rng = np.random.default_rng(6)
    liquid, solid = scene.dam_break_dfsph(5)
    liquid = liquid + rng.uniform(-0.01, 0.01, liquid.shape)
    pos = np.concatenate([liquid, solid]).astype(np.float32)
    o = Oracle(pos, len(liquid), 0.025, mode="clean")
    o.update_grid()
    cells = np.array([o.cell_of(i) for i in range(len(pos))])
    for i in range(0, len(liquid), 5):
        exp = np.nonzero(np.all(np.abs(cells - cells[i]) <= 2, axis=1))[0]
        exp = exp[exp != i]
        got = o.neighbor[i, :o.neighborCount[i]]
        assert sorted(got) == sorted(exp)


def test_cfl_tree_covers_only_a_prefix():
    """Q5: the strided tree leaves max over [0, 2^floor(log2(N-1))) in slot 0."""
    liquid, solid = scene.dam_break_dfsph(5)   # 125 liquid -> prefix 64
    pos = np.concatenate([liquid, solid]).astype(np.float32)
    o = Oracle(pos, 125, 0.025, mode="faithful")
    o.reset_param()
    o.vel[:] = 0
    o.vel[100] = [10, 0, 0]     # outside the prefix
    o.vel[63] = [2, 0, 0]       # last particle inside it
    o.d_vel[:] = 0
    o.cfl_reduce()
    assert o.vel_max[0] == pytest.approx(4.0)
    o.vel[64] = [3, 0, 0]
    o.cfl_reduce()
    assert o.vel_max[0] == pytest.approx(4.0)


def test_time_step_rule():
    """dfsph.py:116-129 on hand-computed numbers."""
    liquid, solid = scene.dam_break_dfsph(5)
    o = Oracle(np.concatenate([liquid, solid]), 125, 0.025)
    o.reset_param()
    o.vel[:] = 0; o.d_vel[:] = 0; o.vel[0] = [1.0, 0, 0]
    o.vs_iter, o.pr_iter = 1, 2
    o.optimize_time_step()      # vmax2 = 1 -> cap = 0.5*0.4*0.05/1 = 0.01 -> clamp 0.005 ; dt = 0.001*1.1
    assert o.deltaT == pytest.approx(0.0011, rel=1e-6)
    o.vs_iter = 11
    o.optimize_time_step()
    assert o.deltaT == pytest.approx(0.0011 * 0.9, rel=1e-6)
    o.vel[0] = [100.0, 0, 0]; o.vs_iter = 7
    o.optimize_time_step()      # cap = 0.01/100 = 1e-4 (== user_min_t)
    assert o.deltaT == pytest.approx(1e-4, rel=1e-6)


def test_pressure_forces_conserve_momentum_without_walls():
    """Pair symmetry: with no boundary the liquid-liquid velocity update of pressure_iter sums to ~0."""
    rng = np.random.default_rng(8)
    liquid = scene.dam_break_liquid(6) + rng.uniform(-0.01, 0.01, (216, 3))
    far = np.array([[-3.0, -3.0, -3.0], [3.0, 3.0, 3.0]])   # two far solids only fix the box
    o = Oracle(np.concatenate([liquid, far]), 216, 0.025, mode="clean")
    o.reset_param(); o.update_grid(); o.compute_density(); o.compute_dfsph_coff()
    o.vel[:] = rng.standard_normal((216, 3)) * 0.1
    o.begin_pressure_iter()
    v0 = o.vel.sum(0).copy()
    o.pressure_iter()
    assert np.abs(o.vel.sum(0) - v0).max() < 1e-4 * np.abs(o.vel).sum()
