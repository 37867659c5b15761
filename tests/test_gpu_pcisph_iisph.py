"""GPU parity of the pcisph.py / iisph.py kernels on a compressed block (spacing 0.92 x 2R,
so that rho > rho0 and the pressure paths are exercised), both neighbour modes, against the oracle."""
import numpy as np
import pytest

from util import rel_err

pytestmark = pytest.mark.gpu
MODES = ["faithful", "clean"]


def _scene(n=9, squeeze=0.92, seed=4):
    """compressed block inside a finely sampled box: ~5.8 k buckets keep the reference's raw list (cells are 0.1
    wide here, Q11: ~10 particles each) below its 2048-entry cap also in faithful mode"""
    from wcsph_amd import scene
    g = (np.arange(n) - n / 2) * 0.05 * squeeze
    liquid = np.stack(np.meshgrid(g, g + 0.3, g, indexing="ij"), -1).reshape(-1, 3)
    liquid = liquid + np.random.default_rng(seed).uniform(-0.003, 0.003, liquid.shape)
    return liquid, scene.box_shell(0.5, 0.035)


@pytest.mark.parametrize("mode", MODES)
def test_pcisph_steps(mode):
    from oracle import Oracle
    from wcsph_amd import pcisph
    liquid, solid = _scene()
    pd = pcisph.init_particle(neighbor_mode=mode, liquid=liquid, solid=solid)
    o = Oracle(pd.pos.to_numpy(), pd.liquid_count, pcisph.gridR, mode=mode)
    o.pcisph_init(pcisph.particleRadius)
    assert abs(pcisph.pci_coff - o.pci_coff) <= 1e-6 * o.pci_coff
    pcisph.reset_param(); o.pcisph_reset_param()
    for s in range(3):
        pd.hash_grid.update_grid(); o.update_grid()
        assert o.overflow_neighbor == 0 and o.overflow_grid == 0
        pcisph.compute_nonpressure_force(); o.pcisph_compute_nonpressure_force()
        assert np.abs(pcisph.rho.to_numpy() - o.rho).max() / o.rho.max() < 1e-5
        if s == 0:
            assert o.rho.max() > 1050.0                            # compressed: the pressure path is live
        assert rel_err(pcisph.d_vel.to_numpy(), o.d_vel) < 1e-4
        pcisph.sovel_pressure(); o.pcisph_solve_pressure()
        assert pcisph.pr_iter == o.pr_iter
        assert rel_err(pcisph.adv_rho.to_numpy(), o.adv_rho) < 1e-5
        assert rel_err(pcisph.pressure.to_numpy(), o.pressure) < 1e-3
        assert rel_err(pcisph.d_vel_pre.to_numpy(), o.d_vel_pre) < 1e-3
        assert abs(pcisph.rho_err[0] - o.rho_err) <= 1e-3 * abs(o.rho_err)
        pcisph.update_pos(); o.pcisph_update_pos()
        assert np.abs(pd.pos.to_numpy() - o.pos).max() < 2e-6
        assert rel_err(pcisph.vel.to_numpy(), o.vel) < 1e-3
    pd.close()


@pytest.mark.parametrize("mode", MODES)
def test_iisph_steps(mode):
    from oracle import Oracle
    from wcsph_amd import iisph
    liquid, solid = _scene()
    pd = iisph.init_particle(neighbor_mode=mode, liquid=liquid, solid=solid)
    o = Oracle(pd.pos.to_numpy(), pd.liquid_count, iisph.gridR, mode=mode)
    o.iisph_init(iisph.particleRadius)
    iisph.reset_param(); o.iisph_reset_param()
    for s in range(3):
        pd.hash_grid.update_grid(); o.update_grid()
        assert o.overflow_neighbor == 0 and o.overflow_grid == 0
        iisph.compute_density(); o.iisph_compute_density()
        assert np.abs(iisph.rho.to_numpy() - o.rho).max() / o.rho.max() < 1e-5
        iisph.compute_nonpressure_force(); o.iisph_compute_nonpressure_force()
        assert iisph.vs_iter == o.vs_iter
        assert rel_err(iisph.d_vel.to_numpy(), o.d_vel) < 1e-3
        iisph.compute_advection(); o.iisph_compute_advection()
        assert rel_err(iisph.d_ii.to_numpy(), o.d_ii) < 1e-4
        assert rel_err(iisph.a_ii.to_numpy(), o.a_ii) < 1e-4
        assert rel_err(iisph.adv_rho.to_numpy(), o.adv_rho) < 1e-5
        iisph.solve_pressure(); o.iisph_solve_pressure()
        assert iisph.pr_iter == o.pr_iter
        assert rel_err(iisph.dij_pj.to_numpy(), o.dij_pj) < 1e-3
        assert rel_err(iisph.pressure.to_numpy(), o.pressure) < 2e-3
        if s == 0:
            assert o.pressure.max() > 0.0
        iisph.update_pos(); o.iisph_update_pos()
        assert np.abs(pd.pos.to_numpy() - o.pos).max() < 2e-6
        assert rel_err(iisph.vel.to_numpy(), o.vel) < 2e-3
    pd.close()
