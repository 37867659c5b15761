"""GPU parity: every dfsph.py kernel of libwcsph_hip.so against the CPU oracle, through the C ABI.

Tolerances: fp32 arithmetic with a different summation order on each side (the reference itself
is order-nondeterministic, Q16).  Per-kernel outputs must agree to TOL = 2e-5 of the field's
max magnitude from identical inputs; the north-star bound is 1e-4 on density."""
import numpy as np
import pytest

from oracle import Oracle
from util import make_pair, rel_err, set_state

pytestmark = pytest.mark.gpu
TOL = 2e-5
MODES = ["faithful", "clean"]


def _random_vel(n, seed=7, scale=0.3):
    return (np.random.default_rng(seed).standard_normal((n, 3)) * scale).astype(np.float32)


@pytest.mark.parametrize("mode", MODES)
def test_neighbor_count_exact(mode):
    g, o = make_pair(8, mode, jitter=0.4)
    g.pd.hash_grid.update_grid()
    o.update_grid()
    np.testing.assert_array_equal(g.pd.hash_grid.neighborCount.to_numpy(), o.neighborCount)
    assert list(g.pd.hash_grid.blockSize.to_numpy()[0]) == list(o.blockSize)
    g.close()


@pytest.mark.parametrize("mode", MODES)
def test_neighbor_rows_are_the_reference_rows(mode):
    """HashGrid.neighbor (ref: HashGrid.py:88-106).  Faithful: the device's rows equal the oracle's ENTRY FOR ENTRY --
    walked cells x outermost, bucket slots in ascending id, duplicates from aliased cells and far aliased particles
    included (they are read from the reference bucket table the library materialises on demand, csrc/bucket.h).
    Clean: the particles of the 5x5x5 walked cells, as a multiset."""
    g, o = make_pair(6, mode, jitter=0.4)
    g.pd.hash_grid.update_grid()
    o.update_grid()
    counts, lists = g.pd.hash_grid.neighbor.rows(0, g.pd.liquid_count, cap=2048)
    np.testing.assert_array_equal(counts, o.neighborCount)
    far = 0
    cells = np.array([o.cell_of(i) for i in range(o.count)])
    for i in range(0, g.pd.liquid_count, 3):
        ref = o.neighbor[i, :o.neighborCount[i]]
        if mode == "faithful":
            np.testing.assert_array_equal(lists[i][:counts[i]], ref)
            far += int((~np.all(np.abs(cells[ref] - cells[i]) <= 2, axis=1)).sum())
        else:
            np.testing.assert_array_equal(np.sort(lists[i][lists[i] >= 0]), np.sort(ref))
    if mode == "faithful":
        assert far > 0, "the scene must contain far aliased entries for this test to mean anything"
    g.close()


def test_reference_bucket_table():
    """HashGrid.gridCount / HashGrid.grid (ref: HashGrid.py:20-21,67-76) read back from the device in faithful mode."""
    g, o = make_pair(6, "faithful", jitter=0.4)
    g.pd.hash_grid.update_grid()
    o.update_grid()
    np.testing.assert_array_equal(g.pd.hash_grid.gridCount, o.gridCount)
    np.testing.assert_array_equal(g.pd.hash_grid.grid.to_numpy(), o.grid)
    h = int(np.argmax(o.gridCount))
    assert g.pd.hash_grid.grid[h, 0] == o.grid[h, 0]
    # after a step the table follows the particles
    g.call("dfsph_reset_param"); o.reset_param()
    g.call("dfsph_step"); o.dfsph_step()
    g.pd.hash_grid.update_grid(); o.update_grid()
    np.testing.assert_array_equal(g.pd.hash_grid.gridCount, o.gridCount)
    np.testing.assert_array_equal(g.pd.hash_grid.grid.to_numpy(), o.grid)
    g.close()


@pytest.mark.parametrize("mode", MODES)
def test_density_and_alpha(mode):
    g, o = make_pair(8, mode, jitter=0.4)
    g.pd.hash_grid.update_grid(); o.update_grid()
    g.call("dfsph_compute_density"); o.compute_density()
    g.call("dfsph_compute_coff"); o.compute_dfsph_coff()
    rho = g.pd.rho.to_numpy()
    assert np.abs(rho - o.rho).max() / o.rho.max() < 1e-5   # north star: <= 1e-4
    assert rel_err(g.alpha_coff.to_numpy(), o.alpha_coff) < TOL
    if mode == "faithful":   # the aliasing must actually matter on this scene, or the test proves nothing
        oc = make_pair(8, "clean", jitter=0.4)
        oc[1].update_grid(); oc[1].compute_density()
        assert np.abs(oc[1].rho - o.rho).max() / o.rho.max() > 1e-3
        oc[0].close()
    g.close()


@pytest.mark.parametrize("mode", MODES)
def test_divergence_kernels(mode):
    g, o = make_pair(8, mode, jitter=0.4)
    nl = g.pd.liquid_count
    for s in (g.call, ):
        pass
    g.call("dfsph_reset_param"); o.reset_param()
    set_state(g, o, vel=_random_vel(nl), kappa_v=np.random.default_rng(3).uniform(-2e-4, 0, nl))
    g.pd.hash_grid.update_grid(); o.update_grid()
    g.call("dfsph_compute_density"); o.compute_density()
    g.call("dfsph_compute_coff"); o.compute_dfsph_coff()
    g.call("dfsph_warmstart_divergence"); o.warmstart_divergence_vel()
    assert rel_err(g.kappa_v.to_numpy(), o.kappa_v) < TOL
    assert rel_err(g.pd.adv_rho.to_numpy(), o.adv_rho) < TOL
    assert rel_err(g.pd.vel.to_numpy(), o.vel) < TOL
    g.call("dfsph_begin_divergence"); o.begin_divergence_iter()
    assert rel_err(g.pd.adv_rho.to_numpy(), o.adv_rho) < TOL
    assert rel_err(g.alpha_coff.to_numpy(), o.alpha_coff) < TOL
    for _ in range(2):
        g.call("dfsph_divergence_iter"); o.divergence_iter()
        assert rel_err(g.pd.vel.to_numpy(), o.vel) < TOL
        assert rel_err(g.pd.adv_rho.to_numpy(), o.adv_rho) < 5 * TOL
        assert abs(g.pd.avg_density_err[0] - o.avg_density_err) <= 1e-4 * abs(o.avg_density_err) + 1e-6
    g.call("dfsph_end_divergence"); o.end_divergence_iter()
    assert rel_err(g.kappa_v.to_numpy(), o.kappa_v) < 5 * TOL
    g.close()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("open_gate", [True, False])
def test_warmstart_pressure_velocity_pass_behind_its_flag(mode, open_gate):
    """warmstart_pressure (dfsph.py:487-508): the velocity pass only acts on particles with adv_rho > rho_L0 (Q7: in the
    reference's units that practically never happens), so the library launches it as a small grid that leaves at once unless
    the elementwise launch before it saw the gate open (flags[FLAG_WARM]).  Here the gate is opened by hand for a third of the
    particles (and left shut): kappa is rescaled either way, the velocities change only behind an open gate, as the oracle's."""
    g, o = make_pair(8, mode, jitter=0.4)
    nl = g.pd.liquid_count
    rng = np.random.default_rng(5)
    g.call("dfsph_reset_param"); o.reset_param()
    set_state(g, o, vel=_random_vel(nl, scale=0.1))
    g.pd.hash_grid.update_grid(); o.update_grid()
    g.call("dfsph_compute_density"); o.compute_density()
    g.call("dfsph_compute_coff"); o.compute_dfsph_coff()
    adv = np.where(rng.random(nl) < 0.33, 2000.0, 0.9).astype(np.float32) if open_gate else np.full(nl, 0.9, np.float32)
    set_state(g, o, adv_rho=adv, kappa=(rng.standard_normal(nl) * 1e-3).astype(np.float32),
              kappa_v=(rng.standard_normal(nl) * 1e-3).astype(np.float32))
    v0 = g.pd.vel.to_numpy().copy()
    for rep in range(2):   # twice: the flag of the first call must not leak into the second (generation numbers)
        g.call("dfsph_warmstart_pressure"); o.warmstart_pressure()
        assert rel_err(g.kappa.to_numpy(), o.kappa) < TOL
        assert rel_err(g.pd.vel.to_numpy(), o.vel) < 2 * TOL
        moved = np.abs(g.pd.vel.to_numpy() - v0).max(axis=1) > 0
        assert moved.any() == open_gate and not moved[adv < 1000.0].any()
        if rep == 0:
            adv = np.full(nl, 0.9, np.float32)             # shut again: the second call must change no velocity
            set_state(g, o, adv_rho=adv)
            v0 = g.pd.vel.to_numpy().copy()
            open_gate = False
    g.close()


@pytest.mark.parametrize("mode", MODES)
def test_pressure_kernels(mode):
    g, o = make_pair(8, mode, jitter=0.4)
    nl = g.pd.liquid_count
    g.call("dfsph_reset_param"); o.reset_param()
    set_state(g, o, vel=_random_vel(nl, scale=0.1))
    g.pd.hash_grid.update_grid(); o.update_grid()
    g.call("dfsph_compute_density"); o.compute_density()
    g.call("dfsph_compute_coff"); o.compute_dfsph_coff()
    g.call("dfsph_warmstart_pressure"); o.warmstart_pressure()
    g.call("dfsph_begin_pressure"); o.begin_pressure_iter()
    assert rel_err(g.pd.adv_rho.to_numpy(), o.adv_rho) < TOL
    for _ in range(3):
        g.call("dfsph_pressure_iter"); o.pressure_iter()
        assert rel_err(g.pd.vel.to_numpy(), o.vel) < 2 * TOL
        assert rel_err(g.pd.adv_rho.to_numpy(), o.adv_rho) < TOL
        assert abs(g.pd.avg_density_err[0] - o.avg_density_err) <= 2e-4 * abs(o.avg_density_err) + 1e-5
    g.call("dfsph_end_pressure"); o.end_pressure_iter()
    assert rel_err(g.kappa.to_numpy(), o.kappa) < 5 * TOL
    g.close()


@pytest.mark.parametrize("mode", MODES)
def test_viscosity_pcg_and_vorticity(mode):
    g, o = make_pair(8, mode, jitter=0.4)
    nl = g.pd.liquid_count
    g.call("dfsph_reset_param"); o.reset_param()
    rng = np.random.default_rng(11)
    set_state(g, o, vel=_random_vel(nl), omega=_random_vel(nl, 5, 0.5),
              vel_guess=(rng.standard_normal((nl, 3)) * 0.01).astype(np.float32))
    g.pd.hash_grid.update_grid(); o.update_grid()
    g.call("dfsph_compute_density"); o.compute_density()
    g.call("dfsph_clear_nonpressure"); o.clear_nonpressure()
    g.call("dfsph_visc_init"); o.init_viscosity_para()
    assert rel_err(g.pd.cg_Minv.to_numpy(), o.cg_Minv) < TOL
    assert rel_err(g.pd.cg_r.to_numpy(), o.cg_r) < 5 * TOL
    assert rel_err(g.pd.cg_dir.to_numpy(), o.cg_dir) < 5 * TOL
    assert abs(g.pd.cg_delta_zero[0] - o.cg_delta_zero) < 1e-4 * abs(o.cg_delta_zero)
    for _ in range(3):
        g.call("dfsph_visc_iter"); o.compute_viscosity_force()
        assert rel_err(g.pd.cg_Ad.to_numpy(), o.cg_Ad) < 1e-4
        assert rel_err(g.pd.vel_guess.to_numpy(), o.vel_guess) < 1e-4
        assert abs(g.pd.cg_delta[0] - o.cg_delta) < 2e-3 * abs(o.cg_delta_zero)
    g.call("dfsph_visc_end"); o.end_viscosity()
    assert rel_err(g.pd.d_vel.to_numpy(), o.d_vel) < 2e-4
    g.call("dfsph_vorticity"); o.compute_vorticity()
    assert rel_err(g.pd.d_omega.to_numpy(), o.d_omega) < 1e-4
    assert rel_err(g.pd.omega.to_numpy(), o.omega) < 1e-4
    assert rel_err(g.pd.d_vel.to_numpy(), o.d_vel) < 2e-4
    g.close()


@pytest.mark.parametrize("mode", MODES)
def test_tension_fixed_semantics(mode):
    g, o = make_pair(8, mode, jitter=0.4, tension_coff=0.5, tension_coff_b=0.5, tension_fixed=1)
    o.tension_coff, o.tension_coff_b, o.tension_fixed = 0.5, 0.5, 1
    g.pd.hash_grid.update_grid(); o.update_grid()
    g.call("dfsph_compute_density"); o.compute_density()
    g.call("dfsph_clear_nonpressure"); o.clear_nonpressure()
    g.call("dfsph_tension"); o.compute_tension()
    assert rel_err(g.pd.normal.to_numpy(), o.normal) < TOL
    assert rel_err(g.pd.d_vel.to_numpy(), o.d_vel) < TOL
    g.close()


@pytest.mark.parametrize("mode", MODES)
def test_cfl_and_time_step(mode):
    g, o = make_pair(8, mode, jitter=0.2)
    nl = g.pd.liquid_count
    g.call("dfsph_reset_param"); o.reset_param()
    v = _random_vel(nl, scale=1.0)
    v[nl - 3] = [30.0, 0, 0]   # beyond the reference tree's prefix (Q5): faithful must NOT see it
    set_state(g, o, vel=v, d_vel=_random_vel(nl, 9, 5.0))
    g.call("dfsph_optimize_time_step")
    if mode == "faithful":
        o.optimize_time_step()
        assert abs(g.pd.vel_max.to_numpy()[0] - o.vel_max[0]) <= 1e-6 * o.vel_max[0]
        assert abs(g.deltaT[0] - o.deltaT) <= 1e-7
        assert g.pd.vel_max.to_numpy()[0] < 100.0
    else:
        assert g.pd.vel_max.to_numpy()[0] > 800.0   # true maximum
    g.close()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("driver", ["fused", "reference_loop"])
def test_full_steps(mode, driver):
    """Three whole DFSPH steps from rest (jittered lattice): fields and iteration counts track the oracle."""
    from wcsph_amd import dfsph
    import util
    liquid, solid = util.small_scene(8, jitter=0.2)
    pd = dfsph.init_particle(particleDim=8, neighbor_mode=mode, boundary=solid)
    pd.pos.from_numpy(np.concatenate([liquid, solid]).astype(np.float32))
    from oracle import Oracle
    o = Oracle(pd.pos.to_numpy(), pd.liquid_count, 0.025, mode=mode)
    dfsph.reset_param(); o.reset_param()
    for s in range(3):
        (dfsph.step if driver == "fused" else dfsph.step_reference_loop)()
        o.dfsph_step()
        assert (dfsph.vs_iter, dfsph.dv_iter, dfsph.pr_iter) == (o.vs_iter, o.dv_iter, o.pr_iter)
        assert abs(dfsph.deltaT[0] - o.deltaT) <= 1e-6 * o.deltaT
        assert np.abs(pd.rho.to_numpy() - o.rho).max() / o.rho.max() < 1e-4
        assert rel_err(pd.vel.to_numpy(), o.vel) < 1e-3
        assert np.abs(pd.pos.to_numpy() - o.pos).max() < 1e-5
        assert rel_err(dfsph.kappa.to_numpy(), o.kappa) < 1e-3
    pd.close()


def test_field_roundtrip_and_order():
    g, o = make_pair(6, "clean", jitter=0.4)
    nl = g.pd.liquid_count
    v = _random_vel(nl)
    g.pd.vel.from_numpy(v)
    g.pd.hash_grid.update_grid()          # re-sorts the device arrays
    np.testing.assert_array_equal(g.pd.vel.to_numpy(), v)   # host view keeps the caller's order, bit exact
    p = g.pd.pos.to_numpy()
    np.testing.assert_array_equal(p, o.pos)
    assert g.pd.vel[5].tolist() == v[5].tolist()
    g.pd.vel[5] = [1.0, 2.0, 3.0]
    assert g.pd.vel.to_numpy()[5].tolist() == [1.0, 2.0, 3.0]
    g.close()


def test_bucket_overflow_is_an_error():
    """ref: HashGrid.py:72-74 drops particles beyond maxInGrid; the library refuses instead (-3)."""
    from wcsph_amd._lib import WcsphError
    liquid = np.tile(np.array([[0.01, 0.21, 0.01]]), (10, 1)) + np.arange(10)[:, None] * 1e-4
    solid = np.array([[-0.5, 0.0, -0.5], [0.5, 1.0, 0.5]])
    import util
    g = util.GpuSim(liquid, solid, mode="clean", maxInGrid=4)
    g.pd.hash_grid.update_grid()
    with pytest.raises(WcsphError) as e:
        g.pd.sync()
    assert e.value.code == -3
    g.close()


def test_escaped_particle_has_no_neighbours():
    """Q4: a particle outside the initial bbox is neither inserted nor given neighbours."""
    g, o = make_pair(6, "faithful", jitter=0.0)
    p = g.pd.pos.to_numpy()
    p[3] = [5.0, 5.0, 5.0]
    g.pd.pos.from_numpy(p); o.pos[3] = p[3]
    g.pd.hash_grid.update_grid(); o.update_grid()
    g.call("dfsph_compute_density"); o.compute_density()
    assert g.pd.hash_grid.neighborCount.to_numpy()[3] == 0 == o.neighborCount[3]
    assert np.abs(g.pd.rho.to_numpy() - o.rho).max() / o.rho.max() < 1e-5
    g.close()


def test_packed_density_records_follow_host_writes():
    """The viscosity / vorticity walks read rho_j from a packed (pos, rho) record written by the density pass
    (csrc/dfsph_kernels.h k_pack_pr): a density written from the host afterwards must reach them."""
    g, o = make_pair(8, "faithful", jitter=0.4)
    nl = g.pd.liquid_count
    g.call("dfsph_reset_param"); o.reset_param()
    rng = np.random.default_rng(5)
    set_state(g, o, vel=_random_vel(nl), omega=_random_vel(nl, 5, 0.5),
              vel_guess=(rng.standard_normal((nl, 3)) * 0.01).astype(np.float32))
    g.pd.hash_grid.update_grid(); o.update_grid()
    g.call("dfsph_compute_density"); o.compute_density()
    rho = (o.rho * (1.0 + 0.05 * rng.random(nl))).astype(np.float32)
    g.pd.rho.from_numpy(rho); o.rho[:] = rho
    g.call("dfsph_clear_nonpressure"); o.clear_nonpressure()
    g.call("dfsph_visc_init"); o.init_viscosity_para()
    assert rel_err(g.pd.cg_Minv.to_numpy(), o.cg_Minv) < TOL
    assert rel_err(g.pd.cg_r.to_numpy(), o.cg_r) < 5 * TOL
    g.call("dfsph_visc_iter"); o.compute_viscosity_force()
    assert rel_err(g.pd.cg_Ad.to_numpy(), o.cg_Ad) < 1e-4
    g.call("dfsph_vorticity"); o.compute_vorticity()
    assert rel_err(g.pd.d_omega.to_numpy(), o.d_omega) < 1e-4
    g.close()


@pytest.mark.parametrize("mode", MODES)
def test_neighbor_count_of_scattered_clusters(mode):
    """Liquid in separate clumps with empty space between them: the faithful neighborCount is computed region by region
    (csrc/grid.h: flags set by k_cell_assign, then k_cellval_occ / k_ncount_box) and must not depend on what lies in between -- exact
    against the oracle, list contents included."""
    from oracle import Oracle
    from util import GpuSim
    rng = np.random.default_rng(3)
    R = 0.025
    centres = np.array([[0.3, 0.4, 0.3], [1.7, 0.5, 0.4], [0.9, 1.6, 1.8], [1.75, 1.7, 1.75], [0.25, 0.3, 1.7], [1.0, 1.0, 1.0]])
    liquid = np.concatenate([c + (rng.random((700, 3)) - 0.5) * 0.38 for c in centres]).astype(np.float32)
    ax = (np.arange(42, dtype=np.float32) + 0.5) * np.float32(2 * R)
    X, Z = np.meshgrid(ax, ax, indexing="ij")
    solid = np.stack([X.ravel(), np.full(X.size, -R, np.float32), Z.ravel()], 1).astype(np.float32)   # a floor: fixes the grid box
    solid = np.concatenate([solid, solid + np.array([0, 2.1 + 2 * R, 0], np.float32)])
    g = GpuSim(liquid, solid, mode=mode)
    o = Oracle(g.pd.pos.to_numpy(), len(liquid), R, mode=mode)
    for step in range(2):
        g.pd.hash_grid.update_grid(); o.update_grid()
        np.testing.assert_array_equal(g.pd.hash_grid.neighborCount.to_numpy(), o.neighborCount)
        g.call("dfsph_compute_density"); o.compute_density()
        assert np.abs(g.pd.rho.to_numpy() - o.rho).max() / o.rho.max() < 1e-5
        pos = o.pos.copy()
        pos[:len(liquid)] += rng.normal(0, 0.02, (len(liquid), 3)).astype(np.float32)   # move every liquid particle a little
        set_state(g, o, pos=pos)
    g.close()


@pytest.mark.parametrize("n,steps,squeeze", [(10, 4, 0.85), (75, 2, 0.93)])
def test_speculative_step_falls_back_when_a_solver_goes_on(n, steps, squeeze):
    """wcsph_dfsph_step queues its common-case continuation behind the solver decisions (taken on the device) -- one
    divergence iteration, one PCG iteration, two pressure iterations.  Here every one of the three decisions falls the OTHER
    way (a squeezed, jittered block with random velocities and a stiff, tightly converged viscosity solve): the gated
    launches must turn out no-ops and the host loops go on exactly as the reference's (dfsph.py:84-103, 131-164).
    n = 10: the whole continuation is queued before the first mail arrives (small problems); n = 75 (421 875 particles): the
    part behind the viscosity decision is queued once the divergence decision is known."""
    from wcsph_amd import dfsph, scene
    from oracle import Oracle
    liquid, solid = scene.dam_break_dfsph(n)
    rng = np.random.default_rng(7)
    c = liquid.mean(0)
    liquid = c + squeeze * (liquid - c) + rng.uniform(-0.2, 0.2, liquid.shape) * scene.PARTICLE_RADIUS
    vel0 = (rng.uniform(-1, 1, (len(liquid), 3)) * 2.0).astype(np.float32)
    pd = dfsph.init_particle(particleDim=n, neighbor_mode="faithful", liquid=liquid, boundary=solid, viscosity=100.0, viscosity_err=1e-4)
    o = Oracle(pd.pos.to_numpy(), pd.liquid_count, 0.025, mode="faithful")
    dfsph.reset_param(); o.reset_param()
    o.viscosity, o.viscosity_err = 100.0, 1e-4
    pd.vel.from_numpy(vel0); o.vel[...] = vel0
    seen = []
    for s in range(steps):
        dfsph.step()
        o.dfsph_step()
        seen.append((o.vs_iter, o.dv_iter, o.pr_iter))
        assert (dfsph.vs_iter, dfsph.dv_iter, dfsph.pr_iter) == seen[-1], (s, seen)
        # dt is CFL-limited here: it follows max |v|, which agrees to ~4e-6 in this violent scene (the per-kernel entry
        # points, which do not speculate, show the same 1-2e-6 in dt)
        assert abs(dfsph.deltaT[0] - o.deltaT) <= 2e-5 * o.deltaT
        assert np.abs(pd.rho.to_numpy() - o.rho).max() / o.rho.max() < 1e-4
        # fields: 1e-3 of the field's scale for (all but a handful of) the particles.  The reference's warm start has a
        # discontinuous gate (`if adv_rho[i] > 0.0`, dfsph.py:421: the whole kick or none, see test_gpu_developed_state.py):
        # the order of the list entries comes from the sort's atomics, and in about one run in four ONE particle of this violent
        # scene flips (2.4 % of the fastest particle's speed) and drags its neighbourhood along in the following step
        nl = pd.liquid_count
        for name, x, y in (("vel", pd.vel.to_numpy(), o.vel), ("pos", pd.pos.to_numpy()[:nl], o.pos[:nl]),
                           ("kappa", dfsph.kappa.to_numpy(), o.kappa), ("vel_guess", pd.vel_guess.to_numpy(), o.vel_guess)):
            dd = np.abs(np.asarray(x, np.float64) - y).reshape(len(x), -1).max(axis=1)
            scale = 1e-2 if name == "pos" else max(np.abs(y).max(), 1e-30)     # pos: 1e-5 absolute
            assert (dd > 1e-3 * scale).sum() <= 3 + 40 * (s + 1), (name, s, int((dd > 1e-3 * scale).sum()))
            assert dd.max() < (0.1 * scale if name != "pos" else 1e-3), (name, s, dd.max())
    a = np.array(seen)
    assert a[:, 0].max() >= 2 and a[:, 1].max() >= 2 and a[:, 2].max() >= 3, seen   # every fallback was taken
    pd.close(); o.close()


def _rebuilt_counts(g, o):
    """gridCount / neighborCount / overflow of the oracle's update_grid() on the library's current positions (a rebuild by
    definition; the SAME oracle throughout: the grid box is that of the initial positions, as the library's)"""
    o.pos[...] = g.pd.pos.to_numpy()
    before = o.overflow_grid
    o.update_grid()
    return np.array(o.gridCount), np.array(o.neighborCount), o.overflow_grid - before


def test_bucket_table_updated_in_place_equals_a_rebuilt_one():
    """One faithful context between two of its own grid updates: k_cell_assign moves the counts of the particles that changed
    their cell (old cell: pos.w) instead of rebuilding the reference's bucket table (api_grid.h grid_assign).  A violent scene
    (random velocities: a tenth of the particles change cells per step), 40 fused steps and an explicit update_grid() every
    fourth step: gridCount (expanded from the byte table on demand) and neighborCount equal those of an oracle freshly built
    on the same positions, every time; a position upload in between (which falls back to a rebuild) changes nothing."""
    from wcsph_amd import dfsph
    g, o = make_pair(10, "faithful", jitter=0.4)
    nl = g.pd.liquid_count
    g.call("dfsph_reset_param")
    g.pd.vel.from_numpy(_random_vel(nl, scale=1.5))
    changed = 0
    prev = None
    for s in range(40):
        g.call("dfsph_step")
        if s % 4 == 3:
            g.pd.hash_grid.update_grid()
            gc, nc, ovf = _rebuilt_counts(g, o)
            assert ovf == 0
            np.testing.assert_array_equal(g.pd.hash_grid.gridCount, gc, err_msg="step %d" % s)
            np.testing.assert_array_equal(g.pd.hash_grid.neighborCount.to_numpy()[:nl], nc[:nl], err_msg="step %d" % s)
            if prev is not None:
                changed += int((gc != prev).sum())
            prev = gc
        if s == 21:   # a position upload: the next update rebuilds, the ones after it update in place again
            p = g.pd.pos.to_numpy(); p[:nl] += np.float32(1e-3); g.pd.pos.from_numpy(p)
    assert changed > 100          # the table did change under the test
    assert g.scalar("table_updates_in_place") >= 46     # 40 steps + 10 explicit updates, less the first and the rebuild after the upload
    g.close(); o.close()


def test_bucket_cap_with_the_table_updated_in_place():
    """The reference's cap (HashGrid.py:72-74, here an error) while the table is updated in place: a bucket AT the cap that one
    particle leaves and another enters in the same update is not an error (whichever atomic lands first), one particle more is.
    Particles are moved by update_pos (a position upload would make the next update a rebuild).  The table has one bucket per
    particle, so cells alias: the layout is searched (on the oracle) for one whose four cells have buckets of their own."""
    from wcsph_amd._lib import WcsphError
    import util
    R, CAP = 0.025, 4
    zs = np.arange(50, dtype=np.float32) * np.float32(2 * R)
    solid = np.concatenate([np.stack([np.full(50, 1.9, np.float32), np.zeros(50, np.float32), zs], 1), [[0.0, 0.5, 0.0]]]).astype(np.float32)

    def counts(liq):
        o2 = Oracle(np.concatenate([liq, solid]).astype(np.float32), len(liq), R, mode="faithful", maxInGrid=CAP)
        o2.update_grid()
        out = (np.array(o2.gridCount), o2.overflow_grid)
        o2.close()
        return out

    for shift in range(40):
        cell = lambda k: np.array([0.07 + (0.25 + 0.05 * shift) * k / 3.0, 0.17, 0.07 + 0.05 * (shift % 7)])   # noqa: E731
        liquid = np.array([cell(0) + [0.002 * t, 0, 0] for t in range(4)] + [cell(1), cell(2)], np.float32)
        step1 = liquid.copy(); step1[0] = cell(3); step1[4] = cell(0) + [0.004, 0.002, 0]
        step2 = step1.copy(); step2[5] = cell(0) + [0.006, 0.004, 0]
        (g0, o0), (g1, o1), (g2, o2) = counts(liquid), counts(step1), counts(step2)
        if o0 == 0 and g0.max() == CAP and o1 == 0 and g1.max() == CAP and o2 > 0:
            break
    else:
        pytest.fail("no layout with four un-aliased cells found")
    g = util.GpuSim(liquid, solid, mode="faithful", maxInGrid=CAP)
    nl = len(liquid)
    g.pd.hash_grid.update_grid(); g.pd.sync()
    np.testing.assert_array_equal(g.pd.hash_grid.gridCount, g0)

    def move(target):
        p = g.pd.pos.to_numpy()[:nl]
        g.pd.vel.from_numpy(((target - p) / np.float32(0.01)).astype(np.float32))
        g.deltaT[0] = 0.01
        g.call("dfsph_update_pos")
        g.pd.hash_grid.update_grid()

    move(step1)                 # particle 0 leaves A for D, particle 4 (from B) enters A: A stays at the cap
    g.pd.sync()
    gc, ovf = counts(g.pd.pos.to_numpy()[:nl])
    assert ovf == 0 and gc.max() == CAP
    np.testing.assert_array_equal(g.pd.hash_grid.gridCount, gc)
    assert g.scalar("table_updates_in_place") == 1
    move(step2)                 # particle 5 (from C) enters A as well: five in a bucket of four
    with pytest.raises(WcsphError) as e:
        g.pd.sync()
    assert e.value.code == -3
    g.close()


@pytest.mark.parametrize("how", ["rebuild_clamps", "in_place_saturates"])
def test_bucket_table_heals_after_a_cap_error(how):
    """the byte table of the bucket occupancy must not stay wrong after a cap error that the caller steps past
    (include/wcsph.h allows that).  rebuild_clamps: 300 particles in one cell at the first update -- k_bucket_sum clamps the
    byte at 255 and reports -3.  in_place_saturates: 60 in the cell, then 200 more arrive by update_pos -- the in-place
    update (k_cell_assign) saturates the byte at 255 instead of carrying into the next bucket's count, and reports -3.
    Either way the update after the report REBUILDS the table: once the crowd has left, gridCount and neighborCount equal
    those of an oracle built on the same positions (ref: HashGrid.py:67-76)."""
    from wcsph_amd._lib import WcsphError
    import util
    R = 0.025
    solid = np.array([[-0.1, -0.1, -0.1], [2.0, 2.0, 2.0]], np.float32)
    n = 300
    k = np.arange(n)
    site = np.stack([0.12 + 0.1 * (k % 15), 0.12 + 0.1 * ((k // 15) % 15), 1.02 + 0.1 * (k // 225)], 1).astype(np.float32)
    crowd = (np.array([0.52, 0.52, 0.52], np.float32) + 1e-4 * np.stack([k % 7, (k // 7) % 7, k // 49], 1)).astype(np.float32)
    stay = 50
    apart = site.copy(); apart[:stay] = crowd[:stay]              # what is left once the crowd has gone: 50 in the cell (cap 64)
    if how == "rebuild_clamps":
        start, packed = crowd.copy(), None
    else:
        start = site.copy(); start[:60] = crowd[:60]
        packed = site.copy(); packed[:260] = crowd[:260]
    g = util.GpuSim(start, solid, mode="faithful")
    g.pd.hash_grid.update_grid()

    def move(target):
        p = g.pd.pos.to_numpy()[:n]
        g.pd.vel.from_numpy(((target - p) / np.float32(0.01)).astype(np.float32))
        g.deltaT[0] = 0.01
        g.call("dfsph_update_pos")
        g.pd.hash_grid.update_grid()

    if packed is not None:
        g.pd.sync()                                                # 60 in the cell: no error yet
        move(packed)                                               # 260: the byte saturates
        assert g.scalar("table_updates_in_place") == 1
    with pytest.raises(WcsphError) as e:
        g.pd.sync()
    assert e.value.code in (-3, -4)                                # (the 300-entry lists of the crowd trip their own cap as well)
    move(apart)
    for _ in range(3):                                             # every condition is handed over once
        try:
            g.pd.sync()
            break
        except WcsphError:
            g.pd.hash_grid.update_grid()
    pos = g.pd.pos.to_numpy()
    o2 = Oracle(pos, n, R, mode="faithful", bounds=(g.pd.maxboundarynp, g.pd.minboundarynp))
    o2.update_grid()
    assert o2.overflow_grid == 0
    np.testing.assert_array_equal(g.pd.hash_grid.gridCount, o2.gridCount)
    np.testing.assert_array_equal(g.pd.hash_grid.neighborCount.to_numpy()[:n], np.array(o2.neighborCount)[:n])
    move(apart + np.float32(0.003))                                # and the table is updated in place again afterwards
    g.pd.sync()
    o2.pos[:n] = g.pd.pos.to_numpy()[:n]
    o2.update_grid()
    np.testing.assert_array_equal(g.pd.hash_grid.gridCount, o2.gridCount)
    g.close(); o2.close()


def test_neighbor_count_follows_moving_particles_update_after_update():
    """Faithful mode, one context: twelve grid updates in a row on moving particles (positions uploaded: every update rebuilds
    the bucket table), each read back at once: neighborCount and gridCount equal the oracle's every time (a histogram / region
    flag left over from the previous update, or -- while rounds 3-4 ran these kernels on a side stream -- a missing join, shows up here)."""
    g, o = make_pair(8, "faithful", jitter=0.4)
    rng = np.random.default_rng(11)
    pos = g.pd.pos.to_numpy().copy()
    nl = g.pd.liquid_count
    for k in range(12):
        pos[:nl] += (rng.standard_normal((nl, 3)) * 0.012).astype(np.float32)
        set_state(g, o, pos=pos)
        g.pd.hash_grid.update_grid()
        o.update_grid()
        np.testing.assert_array_equal(g.pd.hash_grid.neighborCount.to_numpy(), o.neighborCount, err_msg="update %d" % k)
        if k % 4 == 3:
            np.testing.assert_array_equal(g.pd.hash_grid.gridCount, o.gridCount)
    g.close()
