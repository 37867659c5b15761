"""ParticleData with the reference's constructor, methods, attributes and field names
(ref: ParticleData.py:10-185).  Host-side point accumulation and constants are plain
Python / NumPy; every field lives on the GPU inside a libwcsph_hip.so context in
cell-sorted float4-padded SoA arrays and is presented here in the caller's original
particle order through `Field` proxies (.to_numpy / .from_numpy / [i]).

The marching-cubes grid the reference constructs and allocates unconditionally
(ParticleData.py:29,177,184; ~30 GB at 2 M particles) is constructed here too (`mc_grid`,
wcsph_amd.MarchingCubeGrid.MCGrid) but its device arrays are allocated on first use; the
surface-reconstruction kernels (:187-311) are `compute_color_map / cal_anistropic_kernel`
below (csrc/surface.h).
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from .HashGrid import HashGrid
from .field import Field, ScalarField
from .kernels.CubicKernel import CubicKernel
from .MarchingCubeGrid import MCGrid


class ParticleData:
    def __init__(self, particleR, neighbor_mode="faithful", maxInGrid=64, maxNeighbour=2048, cfl_radius=None):
        self.count = 0
        self.liquid_count = 0
        self.solid_count = 0

        self.rho_L0 = 1000.0
        self.rho_S0 = self.rho_L0
        self.VL0 = particleR * particleR * particleR * 0.8 * 8.0
        self.VS0 = self.VL0
        self.liqiudMass = self.VL0 * self.rho_L0

        self.particleR = particleR
        self.cfl_radius = particleR if cfl_radius is None else cfl_radius
        self.neighbor_mode = neighbor_mode
        self.hash_grid = HashGrid(particleR * 2.0, maxInGrid, maxNeighbour, self)
        self.kernel_c = CubicKernel(self.hash_grid.searchR)
        self.mc_grid = MCGrid(particleR, 4, 512, self)   # ref: ParticleData.py:29 (device arrays on first use)

        f = lambda name, nc: Field(self, name, nc)  # noqa: E731
        self.pos = Field(self, "pos", 3, full_count=True)
        self.color = f("color", 1)            # surface reconstruction fields (ParticleData.py:37-42),
        self.color_grad = f("color_grad", 3)  # filled by compute_color_map() / cal_anistropic_kernel()
        self.pos_avr = f("pos_avr", 3)
        self.G = f("G", 9)
        self.normal = f("normal", 3)
        self.vel_guess = f("vel_guess", 3)
        self.vel = f("vel", 3)
        self.omega = f("omega", 3)
        self.vel_max = f("vel_max", 1)
        self.d_vel = f("d_vel", 3)
        self.d_omega = f("d_omega", 3)
        self.pressure = f("pressure", 1)
        self.rho = f("rho", 1)
        self.adv_rho = f("adv_rho", 1)

        # viscosity cg solver
        self.gravity = np.array([0.0, -9.81, 0.0])
        self.dim_coff = 10.0
        self.viscosity = 10.0
        self.viscosity_b = 10.0
        self.viscosity_err = 0.05

        self.avg_density_err = ScalarField(self, "avg_density_err")
        self.cg_delta = ScalarField(self, "cg_delta")
        self.cg_delta_old = ScalarField(self, "cg_delta_old")
        self.cg_delta_zero = ScalarField(self, "cg_delta_zero")

        self.cg_Minv = f("cg_Minv", 9)
        self.cg_r = f("cg_r", 3)
        self.cg_dir = f("cg_dir", 3)
        self.cg_Ad = f("cg_Ad", 3)
        self.cg_s = f("cg_s", 3)

        # tension
        self.tension_coff = 0.0
        self.tension_coff_b = 0.0
        # 0: the reference loop literally (dfsph.py:264-305) over its raw bucket lists -- faithful neighbour mode only;
        # 1: n_i = h * sum (m/rho_j) gradW, curvature term over the support (the documented repair of Q8).
        # None (default): what the neighbour mode implies -- faithful computes what the reference computes (0), clean the
        # repaired form (1).  A slab window (slab.HipBackend) asks for 1: the raw rows span all ranks.
        self.tension_fixed = None

        # vorticity
        self.viscosity_omega = 0.1
        self.vorticity_coff = 0.01
        self.vorticity_init = 0.5

        self.point_list = []
        self._chunks = []  # bulk-added points (add_liquid_points / add_solid_points), kept as arrays

        self.maxboundarynp = np.ones(shape=(1, 3), dtype=np.float32)
        self.minboundarynp = np.ones(shape=(1, 3), dtype=np.float32)
        for j in range(3):
            self.maxboundarynp[0, j] = -10000.0
            self.minboundarynp[0, j] = 10000.0

        # driver-script extras carried through the C ABI (dfsph.py:39-40, sesph.py:23-62)
        self.user_max_t = 0.005
        self.user_min_t = 0.0001
        self.eps = 1e-5
        self.sesph_particle_radius = 0.025
        self.hash_table_size = 0  # 0 => count (HashGrid.py:114)
        self.list_radius = 0.0    # 0 => HashGrid.searchR; sesph.py sets its own (smaller) kernel support
        self.list_cap = 0         # 0 => 256 slots of each kind per particle
        self.ss_VS0_factor = 2.0  # sesph.py:37 / pcisph.py:38 VS0 = 2 VL0 ; iisph.py:37 VS0 = VL0
        self.pci_coff = 0.0       # pcisph.py:299 GetPciCoff()
        self.pc_visc, self.pc_visc_b = 0.05, 0.0              # pcisph.py:68-70
        self.ii_visc, self.ii_visc_b, self.ii_omega = 2.0, 3.0, 0.5   # iisph.py:80-83
        self._ctx = None

    # ---- point accumulation (ref: ParticleData.py:100-138) -------------------------------------
    def _grow_bbox(self, pts):
        pts = np.asarray(pts, dtype=np.float64).reshape(-1, 3)
        if len(pts):
            self.maxboundarynp[0] = np.maximum(self.maxboundarynp[0], pts.max(axis=0).astype(np.float32))
            self.minboundarynp[0] = np.minimum(self.minboundarynp[0], pts.min(axis=0).astype(np.float32))

    def add_liquid_point(self, point):
        if self.solid_count:
            raise ValueError("liquid points must be added before solid points (the kernels test j < liquid_count)")
        self._chunks.append(np.asarray(point, dtype=np.float64).reshape(1, 3))
        self._grow_bbox(point)
        self.count += 1
        self.liquid_count += 1

    def add_solid_point(self, point):
        self._chunks.append(np.asarray(point, dtype=np.float64).reshape(1, 3))
        self._grow_bbox(point)
        self.count += 1
        self.solid_count += 1

    def add_liquid_points(self, pts):
        """Vectorised add_liquid_point for large scenes ([n,3] array)."""
        if self.solid_count:
            raise ValueError("liquid points must be added before solid points")
        pts = np.asarray(pts, dtype=np.float64).reshape(-1, 3)
        self._chunks.append(pts)
        self._grow_bbox(pts)
        self.count += len(pts)
        self.liquid_count += len(pts)

    def add_solid_points(self, pts):
        pts = np.asarray(pts, dtype=np.float64).reshape(-1, 3)
        self._chunks.append(pts)
        self._grow_bbox(pts)
        self.count += len(pts)
        self.solid_count += len(pts)

    def add_obj(self, filename):
        """ref: ParticleData.py:130-138 -- every `v x y z` line becomes a solid point."""
        pts = []
        for line in open(filename, "r"):
            if line.startswith('#'):
                continue
            values = line.split()
            if not values:
                continue
            if values[0] == 'v':
                pts.append(list(map(float, values[1:4])))
        if pts:
            self.add_solid_points(np.array(pts))

    def _initial_positions(self):
        if not self._chunks:
            return np.zeros((0, 3), np.float32)
        return np.ascontiguousarray(np.concatenate(self._chunks, axis=0).astype(np.float32))

    # ---- device set-up (ref: ParticleData.py:142-185) ---------------------------------------------
    def _params(self):
        p = _lib.Params()
        p.n_liquid, p.n_solid = self.liquid_count, self.solid_count
        p.neighbor_mode = _lib.MODE_FAITHFUL if self.neighbor_mode == "faithful" else _lib.MODE_CLEAN
        p.max_in_grid, p.max_neighbour = self.hash_grid.maxInGrid, self.hash_grid.maxNeighbour
        p.hash_table_size = int(self.hash_table_size)
        p.tension_fixed = int(self.tension_fixed) if self.tension_fixed is not None else (0 if self.neighbor_mode == "faithful" else 1)
        p.list_cap = int(self.list_cap)
        p.particle_radius, p.cfl_radius = float(self.particleR), float(self.cfl_radius)
        p.user_max_t, p.user_min_t = float(self.user_max_t), float(self.user_min_t)
        p.rho_L0, p.rho_S0, p.VL0, p.VS0, p.liqiudMass = self.rho_L0, self.rho_S0, self.VL0, self.VS0, self.liqiudMass
        p.gravity[0], p.gravity[1], p.gravity[2] = [float(g) for g in self.gravity]
        p.dim_coff, p.viscosity, p.viscosity_b, p.viscosity_err = self.dim_coff, self.viscosity, self.viscosity_b, self.viscosity_err
        p.tension_coff, p.tension_coff_b = self.tension_coff, self.tension_coff_b
        p.viscosity_omega, p.vorticity_coff, p.vorticity_init = self.viscosity_omega, self.vorticity_coff, self.vorticity_init
        p.eps = self.eps
        # sesph.py:23-62 private constants (pi = 3.1415926)
        r = self.sesph_particle_radius
        gridR = r * 2.0
        sr = gridR * 2.0
        pi = 3.1415926
        h3 = sr * sr * sr
        VL0 = r * r * r * 0.8 * 8.0
        p.ss_searchR, p.ss_mk, p.ss_ml = sr, 8.0 / (pi * h3), 48.0 / (pi * h3)
        p.ss_VL0, p.ss_VS0, p.ss_mass, p.ss_rho0 = VL0, VL0 * self.ss_VS0_factor, VL0 * 1000.0, 1000.0
        p.ss_stiff, p.ss_dim, p.ss_visc, p.ss_visc_b = 50000.0, 10.0, 0.1, 0.0
        p.list_radius = float(self.list_radius)
        p.pci_coff, p.pc_visc, p.pc_visc_b = float(self.pci_coff), self.pc_visc, self.pc_visc_b
        p.ii_visc, p.ii_visc_b, p.ii_omega = self.ii_visc, self.ii_visc_b, self.ii_omega
        return p

    def setup_data_gpu(self):
        """ref: ParticleData.py:142-177 -- allocate every field for `count` / `liquid_count` particles."""
        if self._ctx is not None:
            raise RuntimeError("setup_data_gpu() called twice")
        lib = _lib.load()
        p = self._params()
        ctx = C.c_void_p()
        rc = lib.wcsph_create(C.byref(p), C.byref(ctx))
        self._ctx = ctx
        _lib.check(ctx, rc)
        self.hash_grid.setup_grid_gpu()
        # ref: ParticleData.py:177 -- the MC grid reaches searchR beyond the particles' bounding box
        self.mc_grid.setup_grid_gpu(self.maxboundarynp + np.float32(self.mc_grid.searchR), self.minboundarynp - np.float32(self.mc_grid.searchR))

    def setup_data_cpu(self):
        """ref: ParticleData.py:180-185 -- upload positions, fix the grid box."""
        self.hash_grid.setup_grid_cpu(self.maxboundarynp, self.minboundarynp)
        self.mc_grid.setup_grid_cpu(self.maxboundarynp + np.float32(self.mc_grid.searchR), self.minboundarynp - np.float32(self.mc_grid.searchR))
        print("liqiud particle num:", self.liquid_count, "solid particle num:", self.solid_count)

    # ---- surface reconstruction, per-particle part (ref: ParticleData.py:187-298) ------------------------
    def compute_color_map(self):
        """ref: ParticleData.py:187-215 (needs rho of the current step and a valid grid)"""
        ctx = self._need_ctx()
        _lib.check(ctx, _lib.load().wcsph_surface_color_map(ctx))

    def cal_anistropic_kernel(self):
        """ref: ParticleData.py:220-289: weighted mean positions and the anisotropy matrices G (Yu & Turk)"""
        ctx = self._need_ctx()
        _lib.check(ctx, _lib.load().wcsph_surface_anisotropic_kernel(ctx, float(self.mc_grid.searchR)))

    def export_kernel(self, filename="out/test.obj"):
        """ref: ParticleData.py:302-311 -- debug dump of the colour-field gradient: one `v x y z r g b nx ny nz` line per
        liquid particle (colour = 512 x gradient).  compute_color_map() must have run."""
        import os
        pos = self.pos.to_numpy()[:self.liquid_count]
        grad = self.color_grad.to_numpy()
        d = os.path.dirname(filename)
        if d:
            os.makedirs(d, exist_ok=True)
        np.savetxt(filename, np.hstack([pos, grad * 512.0, grad]), fmt="v" + " %f" * 9)

    def _need_ctx(self):
        if self._ctx is None:
            raise RuntimeError("ParticleData.setup_data_gpu() has not been called")
        return self._ctx

    def sync(self):
        ctx = self._need_ctx()
        _lib.check(ctx, _lib.load().wcsph_sync(ctx))

    def close(self):
        if self._ctx is not None:
            _lib.load().wcsph_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
