"""Boundary sampler: triangle mesh (OBJ) -> Poisson-disk boundary particles (ref: boundry.py:100-460, the offline
tool that produced the reference's model/*_boundry.obj files).  The sampling runs on the GPU
(libwcsph_hip.so: wcsph_sample_surface, csrc/sampler.h); this module is the reference-compatible I/O around it:

    pts = sample_obj("model/box.obj", particleRadius=0.025)      # [n,3] float32
    write_obj("box_boundry.obj", pts)                            # `v x y z` lines, as ParticleData.add_obj reads
"""
import ctypes as C

import numpy as np

from . import _lib

particleRadius = 0.025     # boundry.py:20
trial_total = 10           # boundry.py:417


def loadObj(filename):
    """ref: boundry.py:100-113 -- `v` and `f` lines (a/b/c index forms accepted); returns [n_tri,3,3] float32.
    Faces with more than three vertices are fan-triangulated (the reference assumes triangles)."""
    vertices, tris = [], []
    for line in open(filename, "r"):
        if line.startswith('#'):
            continue
        values = line.split()
        if not values:
            continue
        if values[0] == 'v':
            vertices.append(list(map(float, values[1:4])))
        elif values[0] == 'f':
            idx = [int(v.split('/')[0]) for v in values[1:]]
            idx = [i - 1 if i > 0 else len(vertices) + i for i in idx]
            for k in range(1, len(idx) - 1):
                tris.append((idx[0], idx[k], idx[k + 1]))
    v = np.asarray(vertices, dtype=np.float32)
    return v[np.asarray(tris, dtype=np.int64)]


def sample_triangles(tri, particleRadius=particleRadius, seed=20240901, trials=trial_total, cap=None):
    """Poisson-disk sample of the surface given as [n_tri,3,3] vertices; returns [n,3] float32 points."""
    tri = np.ascontiguousarray(tri, dtype=np.float32).reshape(-1, 3, 3)
    lib = _lib.load()
    if cap is None:
        a = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1).sum()
        cap = int(3.0 * a / (particleRadius * particleRadius)) + 1024      # hexagonal packing bound is ~1.15 / R^2
    out = np.empty((cap, 3), np.float32)
    n = C.c_int32()
    rc = lib.wcsph_sample_surface(tri.ctypes.data, len(tri), float(particleRadius), int(seed) & 0xffffffff, int(trials),
                                  out.ctypes.data, cap, C.byref(n))
    if rc != 0:
        raise _lib.WcsphError(rc, "wcsph_sample_surface failed")
    if n.value > cap:
        raise RuntimeError("sample capacity %d too small (%d points)" % (cap, n.value))
    return out[:n.value].copy()


def sample_obj(filename, particleRadius=particleRadius, seed=20240901, trials=trial_total):
    return sample_triangles(loadObj(filename), particleRadius, seed, trials)


def write_obj(filename, pts):
    """ref: boundry.py:444-449"""
    with open(filename, "w") as fo:
        for p in pts:
            print("v", p[0], p[1], p[2], file=fo)
