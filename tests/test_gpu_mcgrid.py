"""MCGrid on the GPU (csrc/mcgrid.h) against the CPU restatement (oracle/mc_ref.c) and mesh properties
(ref: MarchingCubeGrid.py:159-184 cal_surface_point, :232-328 marching_cube)."""
import numpy as np
import pytest

import oracle
from util import make_pair

pytestmark = pytest.mark.gpu


def mesh_of(g):
    mc = g.pd.mc_grid
    mc.update_grid(); mc.cal_surface_point(); mc.marching_cube()
    n = int(mc.vertex_count[0])
    assert n % 3 == 0 and n > 0
    return mc, mc.triangle[:n].reshape(-1, 3, 3)


@pytest.mark.parametrize("mode", ["clean", "faithful"])
def test_surface_value_matches_restatement(mode):
    g, o = make_pair(8, mode, jitter=0.4)
    g.pd.hash_grid.update_grid(); o.update_grid()
    g.call("dfsph_compute_density"); o.compute_density()
    mc = g.pd.mc_grid
    mc.cal_surface_point()
    got = mc.surface_value.to_numpy()
    nl = g.pd.liquid_count
    want = oracle.mc_value(o.pos[:nl], o.rho, g.pd.liqiudMass, mc.gridR, mc.searchR, mc.minboundarynp, mc.blocknp)
    assert got.shape == want.shape == (mc.grid_num,)
    assert want.max() > 0.9                              # the block's interior is "inside" (isolevel 0.5)
    assert np.abs(got - want).max() < 2e-5 * want.max()
    assert (got[want == 0] == 0).all()
    g.close()


def test_mesh_vertices_are_the_grid_edge_crossings():
    g, o = make_pair(8, "clean", jitter=0.4)
    g.pd.hash_grid.update_grid()
    g.call("dfsph_compute_density")
    mc, tri = mesh_of(g)
    value = mc.surface_value.to_numpy()
    want = oracle.mc_crossings(value, mc.blocknp, mc.minboundarynp, mc.gridR, mc.isolevel)
    got = np.unique(np.round(tri.reshape(-1, 3).astype(np.float64), 6), axis=0)
    ref = np.unique(np.round(want.astype(np.float64), 6), axis=0)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 2e-6
    g.close()


def test_mesh_triangles_equal_the_restatement():
    """marching_cube on the GPU against oracle/mc_ref.c's restatement of MarchingCubeGrid.py:252-328 driven by the same
    case tables, on the GPU's own node field: the same triangles (as a multiset; the reference's append order is an
    atomic counter's)."""
    from wcsph_amd import _lib
    g, o = make_pair(8, "clean", jitter=0.4)
    g.pd.hash_grid.update_grid()
    g.call("dfsph_compute_density")
    mc, tri = mesh_of(g)
    edge = np.empty(256, np.int32); tt = np.empty(256 * 16, np.int32)
    assert _lib.load().wcsph_mc_tables(edge.ctypes.data, tt.ctypes.data) == 0
    want = oracle.mc_triangles(mc.surface_value.to_numpy(), mc.blocknp, mc.minboundarynp, mc.gridR, mc.isolevel, edge, tt)
    assert tri.shape == want.shape
    key = lambda m: np.sort(np.round(m.reshape(len(m), 9).astype(np.float64), 6).view([("", np.float64)] * 9), axis=0)   # noqa: E731
    a, b = key(tri).view(np.float64).reshape(-1, 9), key(want).view(np.float64).reshape(-1, 9)
    assert np.abs(a - b).max() < 2e-6
    g.close()


def test_mesh_is_closed_oriented_and_encloses_the_block(tmp_path):
    g, o = make_pair(8, "clean", jitter=0.2)
    g.pd.hash_grid.update_grid()
    g.call("dfsph_compute_density")
    mc, tri = mesh_of(g)
    key = np.round(tri.reshape(-1, 3).astype(np.float64) / 1e-6).astype(np.int64)
    _, ids = np.unique(key, axis=0, return_inverse=True)
    t = ids.reshape(-1, 3)
    t = t[(t[:, 0] != t[:, 1]) & (t[:, 1] != t[:, 2]) & (t[:, 0] != t[:, 2])]      # slivers welded to a point/line
    # every directed edge is matched by exactly one opposite edge: closed and consistently oriented
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]])
    fwd = {}
    for a, b in e.tolist():
        fwd[(a, b)] = fwd.get((a, b), 0) + 1
    assert all(v == 1 for v in fwd.values())
    assert all((b, a) in fwd for (a, b) in fwd)
    # signed volume (outward normals) ~ number of inside nodes x node volume
    p = tri.astype(np.float64)
    vol = np.einsum("ij,ij->i", p[:, 0], np.cross(p[:, 1], p[:, 2])).sum() / 6.0
    value = mc.surface_value.to_numpy()
    inside = (value >= mc.isolevel).sum() * mc.gridR ** 3
    assert vol > 0
    assert abs(vol - inside) < 0.25 * inside
    # liquid block: 8^3 particles at pitch 2R -> (0.4)^3 = 0.064 m^3, the iso-surface sits about one R outside it
    assert 0.05 < vol < 0.12
    mc.out_dir = str(tmp_path)
    f = mc.export_mesh()
    lines = open(f).read().splitlines()
    assert sum(l.startswith("v ") for l in lines) == len(tri) * 3 and sum(l.startswith("f ") for l in lines) == len(tri)
    mc.frame = 0
    mc.export_surface(0.0)                              # the reference's driver call (MarchingCubeGrid.py:139-156)
    assert mc.frame == 1
    g.close()


def test_truncation_is_reported():
    g, o = make_pair(6, "clean")
    g.pd.hash_grid.update_grid()
    g.call("dfsph_compute_density")
    mc = g.pd.mc_grid
    mc.cal_surface_point()
    mc.marching_cube(max_vertex=30)
    assert mc.vertex_count[0] == 30 and mc.triangle.shape == (30, 3)
    g.close()
