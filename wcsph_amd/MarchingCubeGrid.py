"""MCGrid: surface reconstruction on a dense grid (ref: MarchingCubeGrid.py).  Same constructor, attributes and method
names as the reference's class; the node field and the triangles are computed on the GPU (csrc/mcgrid.h) and
materialised on the host on demand.

    pd.hash_grid.update_grid(); compute_density()      # rho must be current
    pd.mc_grid.cal_surface_point(); pd.mc_grid.marching_cube(); pd.mc_grid.export_mesh()

Differences, on purpose: the particles are not binned a second time (the node sum walks the solver's cell-sorted
particles, so the reference's cap of maxInGrid = 4 particles per MC cell, "mc exceed grid", does not exist), and the
256-case tables are generated instead of read from MCData.txt (see csrc/mcgrid.h)."""
import ctypes as C
import os

import numpy as np

from . import _lib
from .kernels.CubicKernel import CubicKernel

MAX_VERTEX = 3000000   # MarchingCubeGrid.py:8


class MCGrid:
    def __init__(self, particleR, maxInGrid, maxNeighbour, particle_data):
        self.fps = 20.0
        self.frame = 0
        self.maxInGrid = maxInGrid
        self.maxNeighbour = maxNeighbour
        self.particle_data = particle_data
        self.gridR = particleR * 0.9
        self.invGridR = 1.0 / self.gridR
        self.searchR = self.gridR * 4.0
        self.grid_num = 0
        self.isolevel = 0.5
        self.kernel_c = CubicKernel(self.searchR)
        self.maxboundarynp = np.ones(shape=(1, 3), dtype=np.float32)
        self.minboundarynp = np.ones(shape=(1, 3), dtype=np.float32)
        self.blocknp = np.ones(shape=(1, 3), dtype=np.int32)
        self.edgetablenp = np.ones(shape=(256), dtype=np.int32)
        self.tritablenp = np.ones(shape=(256, 16), dtype=np.int32)
        self.vertex_count = np.zeros(1, np.int32)
        self.triangle = np.zeros((0, 3), np.float32)
        self.out_dir = "out"
        self._device_ready = False

    # ref: MarchingCubeGrid.py:60-78
    def setup_grid_gpu(self, maxboundarynp, minboundarynp):
        for i in range(3):
            self.maxboundarynp[0, i] = maxboundarynp[0, i]
            self.minboundarynp[0, i] = minboundarynp[0, i]
        for i in range(3):
            self.blocknp[0, i] = int((float(self.maxboundarynp[0, i]) - float(self.minboundarynp[0, i])) / self.gridR + 1)
        self.grid_num = int(self.blocknp[0, 0]) * int(self.blocknp[0, 1]) * int(self.blocknp[0, 2])
        self._device_ready = False

    # ref: MarchingCubeGrid.py:80-103 (the tables come from the library, not from MCData.txt)
    def setup_grid_cpu(self, maxboundarynp, minboundarynp):
        lib = _lib.load()
        tri = np.empty(256 * 16, np.int32)
        _lib.check(None, lib.wcsph_mc_tables(self.edgetablenp.ctypes.data, tri.ctypes.data))
        self.tritablenp[...] = tri.reshape(256, 16)

    def _ensure_device(self):
        if self._device_ready:
            return
        pd = self.particle_data
        ctx = pd._need_ctx()
        dims = (C.c_int32 * 3)()
        mx = np.ascontiguousarray(self.maxboundarynp.ravel(), np.float32)
        mn = np.ascontiguousarray(self.minboundarynp.ravel(), np.float32)
        _lib.check(ctx, _lib.load().wcsph_mc_setup(ctx, float(self.gridR), float(self.searchR), mx.ctypes.data, mn.ctypes.data, dims))
        assert tuple(dims) == tuple(int(v) for v in self.blocknp[0]), (tuple(dims), self.blocknp)
        self._device_ready = True

    # ref: MarchingCubeGrid.py:141-156 -- nothing to bin here; kept for the reference's call sequence
    def update_grid(self):
        self.vertex_count[0] = 0

    # ref: MarchingCubeGrid.py:159-184
    def cal_surface_point(self):
        self._ensure_device()
        ctx = self.particle_data._need_ctx()
        _lib.check(ctx, _lib.load().wcsph_mc_surface_value(ctx))

    @property
    def surface_value(self):
        return _SurfaceValue(self)

    # ref: MarchingCubeGrid.py:232-328
    def marching_cube(self, max_vertex=MAX_VERTEX):
        self._ensure_device()
        ctx = self.particle_data._need_ctx()
        buf = np.empty((max_vertex, 3), np.float32)
        n = C.c_int32()
        _lib.check(ctx, _lib.load().wcsph_mc_marching_cube(ctx, float(self.isolevel), buf.ctypes.data, max_vertex, C.byref(n)))
        if n.value > max_vertex:
            print("exceed max tri", n.value)
        stored = min(n.value, (max_vertex // 3) * 3)
        self.vertex_count[0] = stored
        self.triangle = buf[:stored].copy()

    # ref: MarchingCubeGrid.py:122-136
    def export_mesh(self):
        os.makedirs(self.out_dir, exist_ok=True)
        filename = os.path.join(self.out_dir, "mc_" + str(self.frame) + ".obj")
        tri_vertex = self.triangle
        n = int(self.vertex_count[0])
        with open(filename, "w") as fo:
            for i in range(n):
                print("v %f %f %f" % (tri_vertex[i, 0], tri_vertex[i, 1], tri_vertex[i, 2]), file=fo)
            for i in range(n // 3):
                print("f %d %d %d" % (3 * i + 1, 3 * i + 2, 3 * i + 3), file=fo)
        return filename

    # ref: MarchingCubeGrid.py:139-156
    def export_surface(self, time):
        time_i = int(time * self.fps)
        if int(time_i) == self.frame:
            self.update_grid()
            self.cal_surface_point()
            self.marching_cube()
            self.export_mesh()
            self.frame += 1


class _SurfaceValue:
    def __init__(self, g):
        self._g = g

    def to_numpy(self):
        g = self._g
        g._ensure_device()
        ctx = g.particle_data._need_ctx()
        out = np.empty(g.grid_num, np.float32)
        _lib.check(ctx, _lib.load().wcsph_mc_value_to_host(ctx, out.ctypes.data, g.grid_num))
        return out
