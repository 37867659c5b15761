"""The generated marching-cubes case tables (csrc/mcgrid.h mc_build_tables, exported by wcsph_mc_tables): pure host
code, checked here without a GPU.  They replace the reference's MCData.txt (MarchingCubeGrid.py:82-94)."""
import itertools

import numpy as np
import pytest

from wcsph_amd import _lib

CORNER = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]])
EDGE = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]   # MarchingCubeGrid.py:276-299


@pytest.fixture(scope="module")
def tables():
    lib = _lib.load()
    edge = np.empty(256, np.int32)
    tri = np.empty(256 * 16, np.int32)
    assert lib.wcsph_mc_tables(edge.ctypes.data, tri.ctypes.data) == 0
    return edge, tri.reshape(256, 16)


def tris_of(row):
    n = int(np.argmax(row < 0)) if (row < 0).any() else len(row)
    assert n % 3 == 0
    return row[:n].reshape(-1, 3)


def edge_mid(e):
    return (CORNER[EDGE[e][0]] + CORNER[EDGE[e][1]]) / 2.0


def test_edge_table_is_the_set_of_straddling_edges(tables):
    edge, _ = tables
    for ci in range(256):
        want = sum(1 << e for e, (a, b) in enumerate(EDGE) if ((ci >> a) & 1) != ((ci >> b) & 1))
        assert edge[ci] == want
    assert edge[0] == 0 and edge[255] == 0 and edge[1] == 0x109 and edge[2] == 0x203   # the two first rows everyone knows


def test_rows_use_exactly_the_crossed_edges(tables):
    edge, tri = tables
    for ci in range(256):
        t = tris_of(tri[ci])
        assert len(t) <= 5
        assert (tri[ci][3 * len(t):] == -1).all()
        used = set(t.ravel().tolist())
        assert used == {e for e in range(12) if (edge[ci] >> e) & 1}
        for a, b, c in t:
            assert len({a, b, c}) == 3


def face_of(e1, e2):
    """axis/side of a cube face containing both edge midpoints, or None"""
    m1, m2 = edge_mid(e1), edge_mid(e2)
    for ax in range(3):
        for side in (0.0, 1.0):
            if m1[ax] == side and m2[ax] == side:
                return ax, side
    return None


def boundary_segments(t):
    cnt = {}
    for a, b, c in t:
        for u, v in ((a, b), (b, c), (c, a)):
            k = (min(u, v), max(u, v))
            cnt[k] = cnt.get(k, 0) + 1
    assert all(v <= 2 for v in cnt.values())
    return {k for k, v in cnt.items() if v == 1}


def test_open_edges_lie_on_cube_faces(tables):
    _, tri = tables
    for ci in range(256):
        for u, v in boundary_segments(tris_of(tri[ci])):
            assert face_of(u, v) is not None, (ci, u, v)


def test_neighbouring_cubes_agree_on_the_shared_face(tables):
    """watertightness: the polygon edges a case leaves on a face equal the ones its neighbour leaves on the same face"""
    _, tri = tables
    segs = [boundary_segments(tris_of(tri[ci])) for ci in range(256)]

    def on_face(ci, ax, side):
        out = set()
        for u, v in segs[ci]:
            if face_of(u, v) == (ax, side) or (edge_mid(u)[ax] == side and edge_mid(v)[ax] == side):
                out.add(frozenset((tuple(edge_mid(u) - side * np.eye(3)[ax]), tuple(edge_mid(v) - side * np.eye(3)[ax]))))
        return out

    for ax in range(3):
        hi = [c for c in range(8) if CORNER[c][ax] == 1]
        lo = [next(d for d in range(8) if (CORNER[d] == CORNER[c] - np.eye(3, dtype=int)[ax]).all()) for c in hi]
        for a in range(256):
            for bits in itertools.product((0, 1), repeat=4):   # free corners of the neighbour
                free = [c for c in range(8) if CORNER[c][ax] == 1]
                b = 0
                for c, d in zip(hi, lo):
                    b |= ((a >> c) & 1) << d
                for c, v in zip(free, bits):
                    b |= v << c
                assert on_face(a, ax, 1.0) == on_face(b, ax, 0.0), (ax, a, b)


def test_normals_point_to_the_flagged_side(tables):
    _, tri = tables
    for c in range(8):                       # one flagged corner: one triangle whose normal points at it
        t = tris_of(tri[1 << c])
        assert len(t) == 1
        p = [edge_mid(e) for e in t[0]]
        n = np.cross(p[1] - p[0], p[2] - p[0])
        assert np.dot(n, CORNER[c] - np.mean(p, axis=0)) > 0
    for c in range(8):                       # complement: normal points away from the only clear corner
        t = tris_of(tri[255 ^ (1 << c)])
        p = [edge_mid(e) for e in t[0]]
        n = np.cross(p[1] - p[0], p[2] - p[0])
        assert np.dot(n, CORNER[c] - np.mean(p, axis=0)) < 0
