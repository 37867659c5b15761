"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the CPU oracle):
CPU: the oracle still reproduces them (regression pin);  GPU: the HIP path reproduces them through the C ABI."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODES = ["faithful", "clean"]


def _load(name):
    return dict(np.load(os.path.join(GOLD, name)))


@pytest.mark.parametrize("mode", MODES)
def test_oracle_reproduces_golden_dfsph(mode):
    from oracle import Oracle
    g = _load("dfsph_7cubed_%s.npz" % mode)
    nl = int(g["n_liquid"])
    o = Oracle(g["pos0"], nl, 0.025, mode=mode)
    o.reset_param(); o.update_grid()
    np.testing.assert_array_equal(o.neighborCount, g["neighborCount0"])
    for s in (1, 2, 3):
        o.dfsph_step()
        assert [o.vs_iter, o.dv_iter, o.pr_iter] == list(g["scalars_%d" % s][1:].astype(int))
        np.testing.assert_allclose(o.deltaT, g["scalars_%d" % s][0], rtol=1e-6)
        np.testing.assert_allclose(o.rho, g["rho_%d" % s], rtol=2e-6)
        np.testing.assert_allclose(o.pos[:nl], g["pos_%d" % s], atol=2e-7)
        np.testing.assert_allclose(o.vel, g["vel_%d" % s], atol=2e-5 * np.abs(g["vel_%d" % s]).max())


@pytest.mark.parametrize("mode", MODES)
def test_oracle_reproduces_golden_sesph(mode):
    from oracle import Oracle
    g = _load("sesph_8cubed_%s.npz" % mode)
    nl = int(g["n_liquid"])
    o = Oracle(g["pos0"], nl, 0.05, mode=mode)
    o.sesph_init(0.025); o.sesph_reset_param()
    for s in (1, 2):
        o.sesph_step()
        np.testing.assert_allclose(o.rho, g["rho_%d" % s], rtol=2e-6)
        np.testing.assert_allclose(o.pos[:nl], g["pos_%d" % s], atol=2e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_hip_reproduces_golden_dfsph(mode):
    from util import GpuSim
    g = _load("dfsph_7cubed_%s.npz" % mode)
    nl = int(g["n_liquid"])
    sim = GpuSim(g["pos0"][:nl], g["pos0"][nl:], mode=mode)
    sim.call("dfsph_reset_param")
    sim.pd.hash_grid.update_grid()
    np.testing.assert_array_equal(sim.pd.hash_grid.neighborCount.to_numpy(), g["neighborCount0"])   # integer: exact
    for s in (1, 2, 3):
        sim.call("dfsph_step")
        assert list(sim.iters()) == list(g["scalars_%d" % s][1:].astype(int))
        assert abs(sim.deltaT[0] - g["scalars_%d" % s][0]) <= 1e-6 * g["scalars_%d" % s][0]
        rho = sim.pd.rho.to_numpy()
        assert np.abs(rho - g["rho_%d" % s]).max() / g["rho_%d" % s].max() < 1e-4      # north-star density bound
        assert np.abs(sim.pd.pos.to_numpy()[:nl] - g["pos_%d" % s]).max() < 1e-5
        assert np.abs(sim.pd.vel.to_numpy() - g["vel_%d" % s]).max() < 1e-3 * np.abs(g["vel_%d" % s]).max()
        assert np.abs(sim.kappa.to_numpy() - g["kappa_%d" % s]).max() <= 1e-3 * np.abs(g["kappa_%d" % s]).max() + 1e-12
    sim.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_hip_reproduces_golden_sesph(mode):
    from util import GpuSim
    g = _load("sesph_8cubed_%s.npz" % mode)
    nl = int(g["n_liquid"])
    sim = GpuSim(g["pos0"][:nl], g["pos0"][nl:], particleR=0.05, mode=mode, sesph_particle_radius=0.025, list_radius=0.1)
    sim.call("sesph_reset_param")
    for s in (1, 2):
        sim.call("sesph_step")
        assert np.abs(sim.pd.rho.to_numpy() - g["rho_%d" % s]).max() / g["rho_%d" % s].max() < 1e-5
        assert np.abs(sim.pd.pos.to_numpy()[:nl] - g["pos_%d" % s]).max() < 1e-6
    sim.close()
