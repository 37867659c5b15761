"""Parity where a production run spends its time: ONE DFSPH step from a state the solver itself produced.

The HIP library runs the dam break until the collapsed column has reached the walls (crowded cells, splashes, neighbour
streams with hop slots in a disordered particle order, list lengths near their caps, a grid box that spans the domain);
its persistent state -- pos, vel, omega, vel_guess, kappa, kappa_v, deltaT, pr_iter, the stale avg_density_err
(ref: dfsph.py:36-41 module globals and the fields that survive update_grid) -- is loaded into the oracle, and both take
one step (ref: dfsph.py:606-617).  Compared: exact raw list lengths (HashGrid.neighborCount), equal iteration counts,
density to the north star's 1e-4, positions to 2e-6 and velocities to 1e-3 m/s (but for at most three warm-start gate flips, see below), every other per-step field to the tolerances of the at-rest tests.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _develop(n, mode, steps):
    from wcsph_amd import dfsph
    pd = dfsph.init_particle(particleDim=n, neighbor_mode=mode)
    dfsph.reset_param()
    L = n * 0.05
    reached = None
    for s in range(steps):
        dfsph.step()
        if reached is None and s % 20 == 19:
            p = pd.pos.to_numpy()[:pd.liquid_count]
            if np.abs(p[:, [0, 2]]).max() > L - 0.06:        # the front has hit a side wall
                reached = s
    return dfsph, pd, reached


@pytest.mark.parametrize("mode", ["faithful", "clean"])
@pytest.mark.parametrize("n,steps", [(40, 600), (48, 450)])
def test_one_step_from_a_developed_state(mode, n, steps):
    from oracle import Oracle
    from wcsph_amd import _lib
    dfsph, pd, reached = _develop(n, mode, steps)
    assert reached is not None, "the front never reached the wall: the state is not developed"
    nl = pd.liquid_count
    ctx, lib = pd._need_ctx(), _lib.load()
    pos = pd.pos.to_numpy()
    # the state really is a developed one: the column has collapsed (centre of mass well below the initial one) and
    # splashed up the walls (particles above the initial column top)
    assert pos[:nl, 1].mean() < 0.6 * (0.2 + 0.5 * n * 0.05)
    state = {f: getattr(pd, f).to_numpy() for f in ("vel", "omega", "vel_guess")}
    state["kappa"], state["kappa_v"] = dfsph.kappa.to_numpy(), dfsph.kappa_v.to_numpy()
    dt0 = float(dfsph.deltaT[0])
    err0 = C.c_float()
    _lib.check(ctx, lib.wcsph_get_scalar(ctx, _lib.SCALAR_ID["avg_density_err"], C.byref(err0)))
    o = Oracle(pos, nl, 0.025, mode=mode, bounds=(pd.maxboundarynp, pd.minboundarynp))
    o.reset_param()
    for f, a in state.items():
        getattr(o, f)[...] = a
    o.deltaT = dt0
    o.avg_density_err = err0.value
    o.vs_iter, o.dv_iter, o.pr_iter = dfsph.vs_iter, dfsph.dv_iter, dfsph.pr_iter
    # one step on both sides
    dt = dfsph.step()
    o.dfsph_step()
    ncg = pd.hash_grid.neighborCount.to_numpy()
    np.testing.assert_array_equal(ncg, o.neighborCount)                       # exact raw list length, every particle
    assert ncg.max() > 1.5 * np.median(ncg[ncg > 0]) or mode == "clean"      # crowded neighbourhoods exist
    assert (dfsph.vs_iter, dfsph.dv_iter, dfsph.pr_iter) == (o.vs_iter, o.dv_iter, o.pr_iter)
    assert abs(dt - o.deltaT) <= 1e-6 * o.deltaT
    rho = pd.rho.to_numpy()
    assert np.abs(rho - o.rho).max() / o.rho.max() < 1e-4                     # north star: per-particle density error
    # velocities and positions: tight for every particle -- but for at most three gate flips and their neighbourhoods.  The reference's warm start
    # kicks particle i with dt * sum (kappa_v_i + kappa_v_j) V gradW only `if adv_rho[i] > 0.0` (dfsph.py:421): a particle
    # whose clipped divergence is zero to rounding gets the whole kick or none, depending on the order its ~35 terms were
    # summed in (the library's list order is the sort's, the oracle's the reference's).  The state under test differs from
    # run to run (the sort's slot order inside a cell comes from atomics), and in about one run in ten ONE particle of
    # 110 k sits on that gate: its velocity then differs by 5e-3 ... 2e-2 (everybody else: ~5e-5), its position by dt
    # times that.  tools/diag_developed_parity.py prints the distribution (56 runs: profiles/r03p_developed_parity.txt).
    v = pd.vel.to_numpy()
    vmax = np.abs(o.vel).max()
    dvel = np.abs(v - o.vel).max(axis=1)
    # at most three flipped particles (>= 5e-3 m/s), each with its neighbourhood feeling it through the solvers (>= 1e-3: a
    # few of its ~35 neighbours), everybody else at rounding level; nobody beyond 2 % of the fastest particle
    assert (dvel >= 5e-3).sum() <= 3, "%d particles beyond 5e-3 m/s" % (dvel >= 5e-3).sum()
    assert (dvel >= 1e-3).sum() <= 3 + 40, "%d particles beyond 1e-3 m/s" % (dvel >= 1e-3).sum()
    assert np.quantile(dvel, 0.999) < 3e-4 and dvel.max() < 2e-2 * vmax
    dpos = np.abs(pd.pos.to_numpy()[:nl] - o.pos[:nl]).max(axis=1)
    assert (dpos >= 2e-6 + 5e-3 * dt).sum() <= 3 and (dpos >= 2e-6).sum() <= 3 + 40, "%d particles beyond 2e-6" % (dpos >= 2e-6).sum()
    assert dpos.max() < 2e-2 * vmax * dt
    for name, a, b in (("kappa", dfsph.kappa.to_numpy(), o.kappa), ("kappa_v", dfsph.kappa_v.to_numpy(), o.kappa_v),
                       ("omega", pd.omega.to_numpy(), o.omega), ("vel_guess", pd.vel_guess.to_numpy(), o.vel_guess)):
        dd = np.abs(a - b).reshape(len(a), -1).max(axis=1)
        tol = 2e-3 * max(np.abs(b).max(), 1e-6)
        assert (dd > tol).sum() <= 3 + 40 and dd.max() <= 20 * tol, name      # the same particles as above
    # the compact streams of the crowded state decode to the in-support part of the oracle's rows (a sample of rows)
    rng = np.random.default_rng(5)
    for i in rng.integers(0, nl, 64):
        cnt, lst = pd.hash_grid.neighbor.compact_rows(int(i), 1, cap=512)
        got = np.sort(lst[0, :cnt[0, 0] + cnt[0, 1]])
        row = o.neighbor[i, :o.neighborCount[i]]
        # in-support w.r.t. the positions the lists were BUILT from (this step's update_grid ran before update_pos)
        d = np.linalg.norm(pos[row] - pos[i], axis=1)
        exp = np.sort(row[d <= 0.1 * (1 + 1e-5)])
        edge = row[np.abs(d - 0.1) < 2e-6]                                    # entries on the support boundary may fall either way
        assert set(got) - set(edge) == set(exp) - set(edge)
    pd.close(); o.close()
