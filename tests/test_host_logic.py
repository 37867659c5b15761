"""Host-side pieces that need no GPU: reference-named constants, point accumulation / bbox, the OBJ
point-cloud loader, scene recipes, host kernel classes vs the oracle's kernels."""
import numpy as np
import pytest

from oracle import Oracle
from wcsph_amd import ParticleData, scene
from wcsph_amd.kernels import AdhesionKernel, CohesionKernel, CubicKernel


def test_particle_data_constants_match_reference():
    pd = ParticleData(0.025)   # ref: ParticleData.py:12-31,61-87
    assert pd.rho_L0 == 1000.0 and pd.rho_S0 == 1000.0
    assert pd.VL0 == pytest.approx(0.025 ** 3 * 0.8 * 8.0) and pd.VS0 == pd.VL0
    assert pd.liqiudMass == pytest.approx(pd.VL0 * 1000.0)
    assert (pd.dim_coff, pd.viscosity, pd.viscosity_b, pd.viscosity_err) == (10.0, 10.0, 10.0, 0.05)
    assert (pd.tension_coff, pd.tension_coff_b) == (0.0, 0.0)
    assert (pd.viscosity_omega, pd.vorticity_coff, pd.vorticity_init) == (0.1, 0.01, 0.5)
    hg = pd.hash_grid
    assert (hg.gridR, hg.searchR, hg.invGridR) == (0.05, 0.1, 1.0 / 0.05)
    assert (hg.maxInGrid, hg.maxNeighbour) == (64, 2048)
    assert pd.kernel_c.searchR == hg.searchR
    for name in ("pos vel vel_guess d_vel omega d_omega normal rho adv_rho pressure vel_max cg_Minv cg_r cg_dir "
                 "cg_Ad cg_s avg_density_err cg_delta cg_delta_old cg_delta_zero").split():
        assert hasattr(pd, name), name


def test_point_accumulation_and_bbox(tmp_path):
    pd = ParticleData(0.025)
    pd.add_liquid_point([0.1, 0.2, 0.3])
    pd.add_liquid_points(np.array([[0.5, -0.2, 0.0], [0.0, 0.9, -0.7]]))
    obj = tmp_path / "b.obj"
    obj.write_text("# comment\nv 1.0 2.0 3.0\n\nvn 0 0 1\nv -1.5 0 0 1.0\nf 1 2 3\n")
    pd.add_obj(str(obj))       # ref: ParticleData.py:130-138: only `v` lines, first three numbers
    assert (pd.count, pd.liquid_count, pd.solid_count) == (5, 3, 2)
    np.testing.assert_allclose(pd.maxboundarynp, [[1.0, 2.0, 3.0]])
    np.testing.assert_allclose(pd.minboundarynp, [[-1.5, -0.2, -0.7]])
    assert pd.maxboundarynp.dtype == np.float32 and pd.maxboundarynp.shape == (1, 3)
    p = pd._initial_positions()
    assert p.dtype == np.float32 and p.shape == (5, 3)
    np.testing.assert_allclose(p[3], [1, 2, 3])


def test_scene_recipes():
    l, s = scene.dam_break_dfsph(6)
    L = 6 * 0.05
    assert l.shape == (216, 3)
    assert s[:, 0].min() == -L and s[:, 0].max() == L and s[:, 1].min() == 0.0 and s[:, 1].max() == 2 * L
    assert len(np.unique(np.round(s, 9), axis=0)) == len(s)          # no duplicated edge points
    on_face = (np.abs(np.abs(s[:, 0]) - L) < 1e-12) | (np.abs(np.abs(s[:, 2]) - L) < 1e-12) | \
              (np.abs(s[:, 1]) < 1e-12) | (np.abs(s[:, 1] - 2 * L) < 1e-12)
    assert on_face.all()
    assert np.all(np.abs(l[:, 0]) < L) and np.all(l[:, 1] > 0)
    assert scene.CONFIGS["C3"] ** 3 == 2000376 and scene.CONFIGS["C2"] ** 3 == 97336


def test_host_kernel_classes_agree_with_the_oracle():
    o = Oracle(np.zeros((8, 3), np.float32) + np.arange(8)[:, None] * 0.01, 4, 0.025, maxNeighbour=8)
    k, coh, adh = CubicKernel(0.1), CohesionKernel(0.1), AdhesionKernel(0.1)
    for r in np.linspace(0.0, 0.12, 25):
        assert float(k.Cubic_W_norm(r)) == pytest.approx(o.W_norm(r), rel=2e-6, abs=1e-3)
        assert float(coh.Cubic_W_norm(r)) == pytest.approx(o.coh_W(r), rel=1e-4, abs=1e-3)
        assert float(adh.Cubic_W_norm(r)) == pytest.approx(o.adh_W(r), rel=1e-4, abs=1e-6)
    rng = np.random.default_rng(0)
    for v in rng.uniform(-0.08, 0.08, (30, 3)):
        np.testing.assert_allclose(k.CubicGradW(v), o.gradW(v), rtol=5e-6, atol=5e-3)
    assert k.m_k == pytest.approx(8 / np.pi) and k.m_l == pytest.approx(48 / np.pi) and k.h3 == pytest.approx(1000.0)


def test_hashgrid_host_hash_matches_oracle():
    from wcsph_amd import HashGrid
    liquid, solid = scene.dam_break_dfsph(5)
    pd = ParticleData(0.025)
    pd.add_liquid_points(liquid); pd.add_solid_points(solid)
    o = Oracle(np.concatenate([liquid, solid]), len(liquid), 0.025)
    cells = np.random.default_rng(2).integers(0, 500, (100, 3))
    got = pd.hash_grid.get_cell_hash(cells)
    for c, h in zip(cells, got):
        assert o.cell_hash(*c) == int(h)
    assert isinstance(pd.hash_grid, HashGrid)


def test_pci_coff_is_the_reference_lattice_sum():
    """pcisph.GetPciCoff (ref: pcisph.py:87-115) as one array expression vs the oracle's loop restatement, and vs a
    hand count: the pitch-2R lattice has 5^3 points in [-h, h]^3, the inner 3^3 = 27 of them strictly inside the support
    (r = 0 included, which contributes no gradient; the six axis points sit exactly ON the support radius)."""
    import oracle.oracle as oo
    from wcsph_amd import pcisph
    got, exp = pcisph.GetPciCoff(), oo._load().orc_pci_coff(0.025)
    assert abs(got - exp) <= 1e-12 * exp
    ax = np.array([-0.1, -0.05, 0.0, 0.05, 0.1])
    g = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    assert int((np.sqrt((g * g).sum(1)) < 0.1 - 1e-12).sum()) == 27


def test_max_of_bit_patterns_is_the_max_of_non_negative_floats():
    """wcsph_slab_callbacks.allreduce_max_bits all-reduces the CFL velocity maximum as an int32 bit pattern (MAX): for
    non-negative IEEE-754 floats, including 0, denormals and +inf, the integer order of the bit patterns is the float order"""
    import numpy as np
    rng = np.random.default_rng(3)
    v = np.concatenate([rng.uniform(0, 1e-3, 50), rng.uniform(0, 1e4, 50), [0.0, 0.1, 1e-45, 3.4e38, np.inf]]).astype(np.float32)
    for _ in range(200):
        pick = rng.choice(v, size=8)
        bits = pick.view(np.int32)
        assert np.array([bits.max()], np.int32).view(np.float32)[0] == pick.max()


def test_oracle_flip_candidate_diagnostic_is_consistent_with_update_drho_divergence():
    """orc_diag_drho_divergence (test infrastructure behind the live-state parity tests, tests/test_gpu_configs.py): S is the
    un-clamped sum of update_drho_divergence (dfsph.py:374-392), so max(S, 0) must be the oracle's own adv_rho wherever the raw
    list length passes the `< 20` test of dfsph.py:391, A >= |S|, and a velocity field without relative motion has S = 0 exactly
    for a particle with liquid neighbours only."""
    import numpy as np
    from oracle import Oracle
    from wcsph_amd import scene
    liquid, solid = scene.dam_break_dfsph(8)
    pos = np.concatenate([liquid, solid]).astype(np.float32)
    o = Oracle(pos, len(liquid), 0.025, mode="clean")
    o.reset_param()
    rng = np.random.default_rng(5)
    o.vel[...] = rng.normal(size=o.vel.shape).astype(np.float32)
    o.update_grid()
    S, A, G = o.diag_drho_divergence()
    o.update_drho_divergence()
    full = np.asarray(o.neighborCount) >= 20
    assert full.sum() > 50
    np.testing.assert_allclose(np.maximum(S[full], 0.0), o.adv_rho[full], rtol=2e-5, atol=2e-5)
    assert (o.adv_rho[~full] == 0).all() and (A >= np.abs(S) - 1e-9).all() and (G > 0).any()
    o.vel[...] = np.float32(1.5)                       # rigid translation: (v_i - v_j) = 0 for every liquid pair
    S2, A2, _ = o.diag_drho_divergence()
    interior = np.ones(len(liquid), bool)
    nb = np.asarray(o.neighbor)[:, :int(np.asarray(o.neighborCount).max())]
    cnt = np.asarray(o.neighborCount)
    has_solid = np.array([(nb[i, :cnt[i]] >= len(liquid)).any() for i in range(len(liquid))])
    assert (~has_solid).sum() > 10 and (S2[~has_solid] == 0).all() and (A2[~has_solid] == 0).all()
    o.close()


def test_bench_lap_groups_cover_every_tag_of_the_fused_step():
    """bench.py's per-group breakdown (config.*.breakdown): every lap tag the library can emit (PASS / ELEM kernel names, the
    grid update's stages, the two solver-iteration scopes, the step boundary) lands in a named group"""
    import bench
    tags = {"grid:cells": "grid_update", "grid:scan+sort": "grid_update", "grid:neighborCount": "grid_update", "k_build_lists": "build_lists",
            "k_density_coff_warm": "density_coff", "k_vel_from_k<1>": "divergence_open", "(k_drho_div<DRHO_BEGIN, true>)": "divergence_open",
            "divergence_iter": "divergence_iter", "pressure_iter": "pressure_iter", "(k_drho_press<DRHO_BEGIN, true>)": "pressure_open",
            "k_warm_press_vel_flagged": "pressure_open", "k_visc_minv_residual": "visc_iter", "k_visc_minv_residual_vort": "visc_iter",
            "k_visc_Ad": "visc_iter", "k_visc_update": "visc_iter", "k_visc_dir": "visc_iter", "k_end_div_begin_nonpressure": "elementwise",
            "k_finish_nonpressure": "elementwise", "k_update_vel_warm_scale": "elementwise", "k_end_press_update_pos": "elementwise",
            "step_boundary": "step_boundary"}
    for tag, group in tags.items():
        assert bench.lap_group(tag) == group, (tag, bench.lap_group(tag))


def test_fd_level_stdout_redirect_keeps_the_json_channel_clean(capfd):
    """bench.py prints ONE JSON line: what C++ layers write to file descriptor 1 while a process group is set up (gloo's
    connection lines, RCCL's banner) must land on stderr"""
    import os
    from wcsph_amd.slab import _fd_stdout_to_stderr
    with _fd_stdout_to_stderr():
        os.write(1, b"[Gloo] Rank 0 is connected to 1 peer ranks\\n")
    os.write(1, b"{}\\n")
    out, err = capfd.readouterr()
    assert out == "{}\\n" and "Gloo" in err


def test_halo_lists_of_a_library_migration_are_materialised_lazily_and_driver_lists_take_over():
    """SlabSim's six halo lists (wcsph_amd/slab.py _HaloList): after a migration inside the library only the counts are known to the
    driver and no torch view exists until somebody asks; assigning a list by hand (the torch path, the tools) switches back to the
    driver's own lists, which the next step hands to the library again."""
    import torch
    from wcsph_amd.slab import SlabSim

    class FakeBackend:
        calls = 0

        def halo_list_views(self):
            FakeBackend.calls += 1
            return torch.arange(10, dtype=torch.int32), torch.arange(100, 107, dtype=torch.int32)

    sim = SlabSim.__new__(SlabSim)
    sim.B = FakeBackend()
    assert sim.send_all is None                                   # nothing set yet
    sim._adopt_library_halo([1234, 6, 4, 3, 4])
    assert sim.n_owned == 1234 and sim._halo_counts() == (10, 7, 6, 3) and FakeBackend.calls == 0
    assert sim.send_l.tolist() == [0, 1, 2, 3, 4, 5] and sim.send_r.tolist() == [6, 7, 8, 9]
    assert sim.ghost_l.tolist() == [100, 101, 102] and sim.ghost_r.tolist() == [103, 104, 105, 106]
    assert sim.send_all.numel() == 10 and sim.ghost_all.numel() == 7 and FakeBackend.calls == 1      # one materialisation serves all six
    assert sim.__dict__["_halo_lib"] is not None                  # reading does not take the lists away from the library
    sim.send_l = torch.zeros(2, dtype=torch.int32)                # a driver that sets its own lists ...
    assert sim.__dict__["_halo_lib"] is None                      # ... owns them from then on (wcsph_slab_set_halo before the step)
    sim.send_r = torch.zeros(1, dtype=torch.int32); sim.send_all = torch.zeros(3, dtype=torch.int32)
    sim.ghost_l = sim.ghost_r = torch.zeros(0, dtype=torch.int32); sim.ghost_all = torch.zeros(0, dtype=torch.int32)
    assert sim._halo_counts() == (3, 0, 2, 0)
