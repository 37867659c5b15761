#!/usr/bin/env python3
"""Wall densities with three boundary samplings of the SAME box [-1,1] x [0,2] x [-1,1] (hygiene): the reference's
shipped model/box_boundry.obj (25 387 points = 1058 / m^2; here from the fixture tests/golden/ref_dfsph_default_scene.npz, which
holds the reference scene's positions as data), the GPU Poisson-disk sampler of wcsph_amd.boundry (csrc/sampler.h: ~984 / m^2)
and the square lattice bench.py's scenes use (scene.box_shell, pitch 0.0307 = 1058 / m^2).
A 20 x 6 x 20 liquid lattice is put on the floor (lowest layer at y = R) and compute_density (dfsph.py:248-262) is run once:
rho of the layers by distance from the wall, and the boundary's own share of it.

    python tools/wall_density.py            (GPU box)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from wcsph_amd import dfsph, boundry, scene  # noqa: E402

R = 0.025
f = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_dfsph_default_scene.npz"))
nl = int(f["n_liquid"])
ref_boundary = f["pos0"][nl:].astype(np.float64)
# the box as 12 triangles for the sampler
c = np.array([[x, y, z] for x in (-1.0, 1.0) for y in (0.0, 2.0) for z in (-1.0, 1.0)])
quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
tri = np.array([[c[a], c[b], c[d]] for a, b, d, e in quads] + [[c[a], c[d], c[e]] for a, b, d, e in quads], np.float32)
gpu_boundary = boundry.sample_triangles(tri, R).astype(np.float64)
lattice_boundary = scene.box_shell(1.0)
# liquid: 20 x 6 x 20 lattice of spacing 2R on the floor
i = np.arange(20)
X, Y, Z = np.meshgrid((i - 10) * 2 * R + R, np.arange(6) * 2 * R + R, (i - 10) * 2 * R + R, indexing="ij")
liquid = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1)
layer = np.rint((liquid[:, 1] - R) / (2 * R)).astype(int)
rows = []
for name, b in (("reference file (model/box_boundry.obj)", ref_boundary), ("GPU Poisson-disk sampler", gpu_boundary), ("square lattice of bench.py's scenes", lattice_boundary)):
    pd = dfsph.init_particle(particleDim=20, neighbor_mode="clean", liquid=liquid, boundary=b)
    dfsph.reset_param()
    pd.hash_grid.update_grid()
    dfsph.compute_density()
    rho = pd.rho.to_numpy()
    pd.close()
    rows.append((name, len(b), len(b) / 24.0, [float(rho[layer == k].mean()) for k in range(4)], [float(rho[layer == k].std()) for k in range(2)]))
base = rows[0][3]
print("| boundary | points | per m^2 | rho layer 0 (y = R) | layer 1 | layer 2 | layer 3 | layer 0 vs reference file | std layer 0 / 1 |")
print("|---|---|---|---|---|---|---|---|---|")
for name, n, dens, rho, sd in rows:
    print("| %s | %d | %.0f | %.2f | %.2f | %.2f | %.2f | %+.2f %% | %.2f / %.2f |" % (name, n, dens, rho[0], rho[1], rho[2], rho[3], 100.0 * (rho[0] / base[0] - 1.0), sd[0], sd[1]))
