"""Deterministic synthetic scenes: the reference's own recipes, vectorised.

dam_break_dfsph(n): ref dfsph.py:59-82 with particleDim = n and the box scaled with n
    (n = 20 is the reference scene; the boundary is generated, not read from
    model/box_boundry.obj: a single-layer shell with ~1058 points/m^2).
sesph_scene(): ref sesph.py:66-92 (8000 liquid + 40^3 lattice shell, 3-D).
"""
import numpy as np

PARTICLE_RADIUS = 0.025
BOUNDARY_PITCH = 0.0307  # square lattice pitch with the point density of model/box_boundry.obj


def dam_break_liquid(nx, ny=None, nz=None, particleRadius=PARTICLE_RADIUS):
    """ref: dfsph.py:69-73: x,z = (i - n/2)*2R + R ; y = j*2R + 0.2 ; index order i = (x*ny + y)*nz + z."""
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    dis = particleRadius * 2.0
    i = np.arange(nx * ny * nz, dtype=np.int64)
    ZxY = nz * ny
    x = (i // ZxY - nx / 2).astype(np.float64) * dis + dis * 0.5
    y = ((i % ZxY) // nz).astype(np.float64) * dis + 0.2
    z = (i % nz - nz / 2).astype(np.float64) * dis + dis * 0.5
    return np.stack([x, y, z], axis=1)


def _face_axis(L, pitch, lo, hi, inclusive):
    n = max(int(round((hi - lo) / pitch)), 1)
    t = lo + (hi - lo) * np.arange(n + 1, dtype=np.float64) / n
    return t if inclusive else t[1:-1]


def box_shell(L, pitch=BOUNDARY_PITCH):
    """Single-layer point shell on the 6 faces of [-L,L] x [0,2L] x [-L,L]; no duplicated edge points."""
    xs_in, xs_ex = _face_axis(L, pitch, -L, L, True), _face_axis(L, pitch, -L, L, False)
    ys_ex = _face_axis(L, pitch, 0.0, 2 * L, False)
    pts = []
    for y in (0.0, 2 * L):  # floor / lid: full square
        X, Z = np.meshgrid(xs_in, xs_in, indexing="ij")
        pts.append(np.stack([X.ravel(), np.full(X.size, y), Z.ravel()], 1))
    for x in (-L, L):  # x walls: y open interval, z full
        Y, Z = np.meshgrid(ys_ex, xs_in, indexing="ij")
        pts.append(np.stack([np.full(Y.size, x), Y.ravel(), Z.ravel()], 1))
    for z in (-L, L):  # z walls: x and y open intervals
        X, Y = np.meshgrid(xs_ex, ys_ex, indexing="ij")
        pts.append(np.stack([X.ravel(), Y.ravel(), np.full(X.size, z)], 1))
    return np.concatenate(pts, 0)


def dam_break_dfsph(n, particleRadius=PARTICLE_RADIUS, pitch=BOUNDARY_PITCH):
    """(liquid[n^3,3], solid[m,3]) float64 arrays of the scaled DFSPH dam break (L = n * 2R)."""
    L = n * particleRadius * 2.0
    return dam_break_liquid(n, particleRadius=particleRadius), box_shell(L, pitch)


def sesph_scene(particleRadius=PARTICLE_RADIUS, dim=20, boundary=2.0):
    """ref: sesph.py:66-92."""
    gridR = particleRadius * 2.0
    invGridR = 1.0 / gridR
    blockSize = int(boundary * invGridR)
    i = np.arange(dim ** 3, dtype=np.int64)
    ZxY = dim * dim
    liquid = np.stack([(i // ZxY).astype(np.float64) * gridR, ((i % ZxY) // dim).astype(np.float64) * gridR - 0.9,
                       (i % dim).astype(np.float64) * gridR], 1)
    A = boundary / (float(blockSize) - 1.0)
    B = -0.5 * boundary
    g = np.arange(blockSize ** 3, dtype=np.int64)
    ix, iy, iz = g // (blockSize * blockSize), (g % (blockSize * blockSize)) // blockSize, g % blockSize
    shell = (ix == 0) | (iy == 0) | (iz == 0) | (ix == blockSize - 1) | (iy == blockSize - 1) | (iz == blockSize - 1)
    solid = np.stack([A * ix[shell].astype(np.float64) + B, A * iy[shell].astype(np.float64) + B,
                      A * iz[shell].astype(np.float64) + B], 1)
    return liquid, solid


ADHESION_CENTRE = (0.0, 0.5, 0.0)   # ref: dfsph.py:300 -- the only boundary particles the adhesion term acts on lie
ADHESION_GATE = 0.26                #                      within 0.26 of this point ("the box particle are not included")


def sphere_shell(centre=ADHESION_CENTRE, radius=0.25, n_points=5447):
    """Fibonacci lattice on a sphere.  Defaults: the ball of the reference's model/sphere_boundry.obj scene (5447 of its
    points lie on a sphere of radius 0.25 about (0, 0.5, 0), exactly the set dfsph.py:300-302 gates the adhesion to)."""
    k = np.arange(n_points, dtype=np.float64) + 0.5
    phi = np.arccos(1.0 - 2.0 * k / n_points)
    theta = np.pi * (1.0 + 5.0 ** 0.5) * k
    d = np.stack([np.cos(theta) * np.sin(phi), np.cos(phi), np.sin(theta) * np.sin(phi)], 1)
    return np.asarray(centre, np.float64) + radius * d


def dam_break_with_ball(n, particleRadius=PARTICLE_RADIUS, pitch=BOUNDARY_PITCH, radius=0.25):
    """The scaled dam break with the adhesion ball of the reference's sphere scene inside the liquid column: lattice
    particles closer than R to the ball's surface or inside it are left out.  (liquid, solid) float64."""
    liquid, shell = dam_break_dfsph(n, particleRadius, pitch)
    d = np.linalg.norm(liquid - np.asarray(ADHESION_CENTRE), axis=1)
    return liquid[d > radius + particleRadius], np.concatenate([shell, sphere_shell(radius=radius)], 0)


# BASELINE.json configs -> particleDim of the DFSPH dam break
CONFIGS = {"C2": 46, "C3": 126, "C4": 252, "C5": 100, "default": 20}
