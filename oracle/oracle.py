"""ctypes front-end of oracle/wcsph_ref.c (TEST INFRASTRUCTURE ONLY, parity unpinned).

The class mirrors the reference objects a driver script touches: the ParticleData /
HashGrid fields (numpy views on the C arrays, original particle order, liquid first) and
the zero-argument step functions of dfsph.py / sesph.py under their reference names.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libwcsph_oracle.so")


def _cpu_tag():
    import hashlib
    try:
        txt = "".join(sorted(set(l for l in open("/proc/cpuinfo") if l.startswith(("model name", "flags")))))
    except OSError:
        txt = "unknown"
    return hashlib.sha1(txt.encode()).hexdigest()[:10]


def build_baseline():
    """gcc -O3 -march=native -fopenmp build of the same restatement for bench.py's cpu_baseline leg, compiled on (and
    named after) the CPU it runs on.  WCSPH_ORACLE_BASELINE=1 makes this process load it instead of the checker build."""
    out = os.path.join(_HERE, "_build", "libwcsph_oracle_baseline_%s.so" % _cpu_tag())
    srcs = [os.path.join(_HERE, f) for f in ("wcsph_ref.c", "poisson_ref.c", "mc_ref.c", "Makefile")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "baseline", "BASE_OUT=" + os.path.relpath(out, _HERE)])
    return out

_SCALARS = ["deltaT", "avg_density_err", "cg_delta", "cg_delta_old", "cg_delta_zero", "viscosity", "viscosity_b",
            "tension_coff", "tension_coff_b", "VL0", "VS0", "liqiudMass", "searchR", "gridR", "invGridR", "rho_L0",
            "viscosity_err", "vorticity_coff", "viscosity_omega", "vorticity_init", "rho_err", "pci_coff"]
_INTS = ["vs_iter", "dv_iter", "pr_iter", "overflow_grid", "overflow_neighbor", "table_size", "bs0", "bs1", "bs2",
         "tension_fixed", "mode", "n_reduce"]
_VEC3 = ["vel", "vel_guess", "d_vel", "omega", "d_omega", "normal", "cg_r", "cg_dir", "cg_Ad", "cg_s",
         "d_vel_pre", "pos_star", "vel_star", "d_ii", "dij_pj", "color_grad", "pos_avr"]
_SCAL = ["rho", "adv_rho", "pressure", "vel_max", "alpha_coff", "kappa", "kappa_v", "kbuf", "a_ii", "pressure_pre", "color"]
_STEPS = ["update_grid", "reset_param", "compute_density", "compute_dfsph_coff", "update_drho_divergence",
          "update_drho_pressure", "warmstart_divergence_vel", "begin_divergence_iter", "divergence_iter",
          "end_divergence_iter", "warmstart_pressure", "begin_pressure_iter", "pressure_iter", "end_pressure_iter",
          "clear_nonpressure", "compute_tension", "init_viscosity_para", "compute_viscosity_force", "end_viscosity",
          "compute_vorticity", "cfl_reduce", "optimize_time_step", "update_vel", "update_pos",
          "compute_nonpressure_force", "solve_vel_divergence", "solve_pressure", "dfsph_step",
          "sesph_reset_param", "sesph_update_advection_density", "sesph_update_pressure", "sesph_compute_force",
          "sesph_integrate", "sesph_step",
          # loop-boundary splits of the fused kernels (used by the slab-decomposition tests)
          "warmstart_divergence_drho", "warmstart_divergence_velloop", "make_k_div", "make_k_press", "vel_iter",
          "drho_div_iter", "drho_press_iter", "warmstart_pressure_scale", "warmstart_pressure_vel", "tension_normal",
          "tension_accel", "visc_guess", "visc_minv", "visc_residual", "visc_Ad", "visc_update", "visc_dir",
          "vorticity_accumulate", "vorticity_integrate",
          # pcisph.py / iisph.py
          "pcisph_reset_param", "pcisph_compute_nonpressure_force", "pcisph_init_iter_info", "pcisph_update_iter_info",
          "pcisph_predict_density", "pcisph_solve_pressure", "pcisph_update_pos", "pcisph_step",
          "iisph_reset_param", "iisph_compute_density", "iisph_init_viscosity_para", "iisph_compute_viscosity_force",
          "iisph_combine_nonpressure", "iisph_compute_nonpressure_force", "iisph_compute_advection",
          "iisph_update_iter_info", "iisph_update_pressure_force", "iisph_solve_pressure", "iisph_update_pos",
          "iisph_step",
          "compute_color_map", "cal_anistropic_kernel"]


def build_oracle(force=False):
    """Compile oracle/wcsph_ref.c with gcc (oracle/Makefile). Returns the .so path."""
    srcs = [os.path.join(_HERE, f) for f in ("wcsph_ref.c", "poisson_ref.c", "mc_ref.c", "Makefile")]
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB


_lib = None


def _load():
    global _lib
    if _lib is None:
        lib = C.CDLL(build_baseline() if os.environ.get("WCSPH_ORACLE_BASELINE") == "1" else build_oracle())
        lib.orc_create.restype = C.c_void_p
        lib.orc_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int]
        lib.orc_destroy.argtypes = [C.c_void_p]
        lib.orc_setup.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.orc_setup.restype = C.c_int
        lib.orc_cell_hash.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        lib.orc_cell_hash.restype = C.c_int
        lib.orc_cell_of.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.orc_W_norm.argtypes = [C.c_void_p, C.c_float]
        lib.orc_W_norm.restype = C.c_float
        lib.orc_coh_W.argtypes = [C.c_void_p, C.c_float]
        lib.orc_coh_W.restype = C.c_float
        lib.orc_adh_W.argtypes = [C.c_void_p, C.c_float]
        lib.orc_adh_W.restype = C.c_float
        lib.orc_gradW.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.orc_sesph_init.argtypes = [C.c_void_p, C.c_double]
        lib.orc_pcisph_init.argtypes = [C.c_void_p, C.c_double]
        lib.orc_iisph_init.argtypes = [C.c_void_p, C.c_double]
        lib.orc_pci_coff.argtypes = [C.c_double]
        lib.orc_pci_coff.restype = C.c_double
        lib.orc_get_scalar.argtypes = [C.c_void_p, C.c_int]
        lib.orc_get_scalar.restype = C.c_float
        lib.orc_set_scalar.argtypes = [C.c_void_p, C.c_int, C.c_float]
        lib.orc_get_int.argtypes = [C.c_void_p, C.c_int]
        lib.orc_get_int.restype = C.c_int
        lib.orc_set_int.argtypes = [C.c_void_p, C.c_int, C.c_int]
        for nm in ("dAd", "dnew"):
            getattr(lib, "orc_get_" + nm).argtypes = [C.c_void_p]
            getattr(lib, "orc_get_" + nm).restype = C.c_double
            getattr(lib, "orc_set_" + nm).argtypes = [C.c_void_p, C.c_double]
        for s in _STEPS:
            getattr(lib, "orc_" + s).argtypes = [C.c_void_p]
            getattr(lib, "orc_" + s).restype = None
        for f in _VEC3 + _SCAL + ["pos", "cg_Minv", "G"]:
            getattr(lib, "orc_ptr_" + f).argtypes = [C.c_void_p]
            getattr(lib, "orc_ptr_" + f).restype = C.POINTER(C.c_float)
        for f in ["gridCount", "grid", "neighborCount", "neighbor"]:
            getattr(lib, "orc_ptr_" + f).argtypes = [C.c_void_p]
            getattr(lib, "orc_ptr_" + f).restype = C.POINTER(C.c_int)
        lib.orc_set_num_threads.argtypes = [C.c_int]
        lib.orc_diag_drho_divergence.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.orc_mc_value.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.orc_mc_crossings.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_float, C.c_void_p, C.c_long]
        lib.orc_mc_crossings.restype = C.c_long
        lib.orc_mc_triangles.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
        lib.orc_mc_triangles.restype = C.c_long
        lib.orc_poisson_sample.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_uint32, C.c_int, C.c_void_p, C.c_int]
        lib.orc_poisson_sample.restype = C.c_int
        lib.orc_ps_distance.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.orc_ps_distance.restype = C.c_float
        lib.orc_get_max_threads.restype = C.c_int
        _lib = lib
    return _lib


def mc_value(pos, rho, mass, grid_r, search_r, minb, dims):
    """MCGrid.cal_surface_point restated (oracle/mc_ref.c): node field [nx*ny*nz] from the liquid particles."""
    pos = np.ascontiguousarray(pos, np.float32); rho = np.ascontiguousarray(rho, np.float32)
    minb = np.ascontiguousarray(minb, np.float32).ravel(); dims = np.ascontiguousarray(dims, np.int32).ravel()
    out = np.empty(int(dims[0]) * int(dims[1]) * int(dims[2]), np.float32)
    _load().orc_mc_value(pos.ctypes.data, rho.ctypes.data, len(pos), float(mass), float(grid_r), float(search_r),
                         minb.ctypes.data, dims.ctypes.data, out.ctypes.data)
    return out


def mc_crossings(value, dims, minb, grid_r, iso):
    """every grid edge whose end values straddle iso, interpolated as MCGrid.vertex_interp does -> [n,3]"""
    value = np.ascontiguousarray(value, np.float32)
    minb = np.ascontiguousarray(minb, np.float32).ravel(); dims = np.ascontiguousarray(dims, np.int32).ravel()
    lib = _load()
    n = lib.orc_mc_crossings(value.ctypes.data, dims.ctypes.data, minb.ctypes.data, float(grid_r), float(iso), None, 0)
    out = np.empty((n, 3), np.float32)
    lib.orc_mc_crossings(value.ctypes.data, dims.ctypes.data, minb.ctypes.data, float(grid_r), float(iso), out.ctypes.data, n)
    return out


def mc_triangles(value, dims, minb, grid_r, iso, edge_table, tri_table):
    """MCGrid.marching_cube restated (oracle/mc_ref.c) with explicit case tables -> [n_tri, 3, 3] vertices"""
    value = np.ascontiguousarray(value, np.float32)
    minb = np.ascontiguousarray(minb, np.float32).ravel(); dims = np.ascontiguousarray(dims, np.int32).ravel()
    et = np.ascontiguousarray(edge_table, np.int32).ravel(); tt = np.ascontiguousarray(tri_table, np.int32).reshape(256, 16)
    lib = _load()
    a = (value.ctypes.data, dims.ctypes.data, minb.ctypes.data, float(grid_r), float(iso), et.ctypes.data, tt.ctypes.data)
    n = lib.orc_mc_triangles(*a, None, 0)
    out = np.empty((n, 3), np.float32)
    lib.orc_mc_triangles(*a, out.ctypes.data, n)
    return out.reshape(-1, 3, 3)


def poisson_sample(tri, radius=0.025, seed=20240901, trials=10, cap=None):
    """CPU restatement of the boundary sampler (oracle/poisson_ref.c); tri: [n_tri,3,3] float32 -> [n,3] float32"""
    tri = np.ascontiguousarray(tri, np.float32).reshape(-1, 9)
    if cap is None:
        t = tri.reshape(-1, 3, 3)
        cap = int(3.0 * 0.5 * np.linalg.norm(np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]), axis=1).sum() / radius ** 2) + 1024
    out = np.empty((cap, 3), np.float32)
    n = _load().orc_poisson_sample(tri.ctypes.data, len(tri), float(radius), int(seed) & 0xffffffff, int(trials), out.ctypes.data, cap)
    return out[:min(n, cap)].copy()


def set_num_threads(n):
    _load().orc_set_num_threads(int(n))


def get_max_threads():
    return int(_load().orc_get_max_threads())


class Oracle:
    """One reference simulation state.

    Oracle(pos, n_liquid, particleR, mode) follows ParticleData(particleR) + add_*_point
    for every row of `pos` (liquid rows first) + setup_data_gpu/cpu
    (ref: ParticleData.py:12-185, HashGrid.py:10-54).
    mode: "faithful" (reference hash table, bucket multiset) or "clean" (collision-free).
    """

    def __init__(self, pos, n_liquid, particleR=0.025, mode="faithful", maxInGrid=64, maxNeighbour=2048,
                 bounds=None):
        lib = _load()
        pos = np.ascontiguousarray(pos, dtype=np.float32)
        n = pos.shape[0]
        d = self.__dict__
        d["lib"] = lib
        d["count"], d["liquid_count"], d["solid_count"] = n, int(n_liquid), n - int(n_liquid)
        d["maxInGrid"], d["maxNeighbour"] = maxInGrid, maxNeighbour
        d["h"] = lib.orc_create(self.liquid_count, self.solid_count, float(particleR), maxInGrid, maxNeighbour,
                                0 if mode == "faithful" else 1)
        if bounds is None:  # running bbox of every added point, f32 (ParticleData.py:100-126)
            maxb, minb = pos.max(axis=0), pos.min(axis=0)
        else:
            maxb, minb = bounds
        d["maxboundarynp"] = np.ascontiguousarray(maxb, dtype=np.float32).reshape(1, 3)
        d["minboundarynp"] = np.ascontiguousarray(minb, dtype=np.float32).reshape(1, 3)
        rc = lib.orc_setup(self.h, pos.ctypes.data, self.maxboundarynp.ctypes.data, self.minboundarynp.ctypes.data)
        if rc != 0:
            raise MemoryError("oracle table allocation failed")
        nl = self.liquid_count
        d["pos"] = self._view("pos", (n, 3))
        for f in _VEC3:
            d[f] = self._view(f, (nl, 3))
        for f in _SCAL:
            d[f] = self._view(f, (nl,))
        d["cg_Minv"] = self._view("cg_Minv", (nl, 3, 3))
        d["G"] = self._view("G", (nl, 3, 3))
        d["neighborCount"] = self._view("neighborCount", (nl,), np.int32)
        d["neighbor"] = self._view("neighbor", (nl, maxNeighbour), np.int32)
        ts = self.table_size
        d["gridCount"] = self._view("gridCount", (ts,), np.int32)
        d["grid"] = self._view("grid", (ts, maxInGrid), np.int32)
        for s in _STEPS:
            d[s] = (lambda fn: (lambda: fn(self.h)))(getattr(lib, "orc_" + s))

    def _view(self, name, shape, dtype=np.float32):
        p = getattr(self.lib, "orc_ptr_" + name)(self.h)
        n = int(np.prod(shape))
        if n == 0:
            return np.zeros(shape, dtype)
        ct = C.c_float if dtype == np.float32 else C.c_int
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(n,)).reshape(shape)

    def __getattr__(self, name):
        if name in _SCALARS:
            return float(self.lib.orc_get_scalar(self.h, _SCALARS.index(name)))
        if name in _INTS:
            return int(self.lib.orc_get_int(self.h, _INTS.index(name)))
        raise AttributeError(name)

    def __setattr__(self, name, v):
        if name in _SCALARS:
            self.lib.orc_set_scalar(self.h, _SCALARS.index(name), float(v))
        elif name in ("vs_iter", "dv_iter", "pr_iter", "tension_fixed", "n_reduce"):
            self.lib.orc_set_int(self.h, _INTS.index(name), int(v))
        elif name in self.__dict__ and isinstance(self.__dict__[name], np.ndarray):
            self.__dict__[name][...] = v  # field assignment copies into the C array
        else:
            object.__setattr__(self, name, v)

    @property
    def blockSize(self):
        return (self.bs0, self.bs1, self.bs2)

    def cell_hash(self, x, y, z):
        return int(self.lib.orc_cell_hash(self.h, int(x), int(y), int(z)))

    def cell_of(self, i):
        out = np.zeros(3, np.int32)
        self.lib.orc_cell_of(self.h, int(i), out.ctypes.data)
        return out

    def W_norm(self, r):
        return float(self.lib.orc_W_norm(self.h, float(r)))

    def gradW(self, r):
        r = np.ascontiguousarray(r, np.float32)
        out = np.zeros(3, np.float32)
        self.lib.orc_gradW(self.h, r.ctypes.data, out.ctypes.data)
        return out

    def coh_W(self, r):
        return float(self.lib.orc_coh_W(self.h, float(r)))

    def adh_W(self, r):
        return float(self.lib.orc_adh_W(self.h, float(r)))

    def diag_drho_divergence(self):
        """(S, A, G) per liquid particle on the current lists / positions / velocities: the un-clamped sum of
        update_drho_divergence (dfsph.py:374-392), sum |term| and sum V |gradW| -- how close a particle sits to the gate
        `if adv_rho[i] > 0` of dfsph.py:421 (test diagnostic; call update_grid() first)"""
        nl = self.liquid_count
        S, A, G = (np.zeros(nl, np.float64) for _ in range(3))
        self.lib.orc_diag_drho_divergence(self.h, S.ctypes.data, A.ctypes.data, G.ctypes.data)
        return S, A, G

    def get_dAd(self): return float(self.lib.orc_get_dAd(self.h))          # noqa: E704
    def set_dAd(self, v): self.lib.orc_set_dAd(self.h, float(v))           # noqa: E704
    def get_dnew(self): return float(self.lib.orc_get_dnew(self.h))        # noqa: E704
    def set_dnew(self, v): self.lib.orc_set_dnew(self.h, float(v))         # noqa: E704

    def sesph_init(self, particleRadius=0.025):
        self.lib.orc_sesph_init(self.h, float(particleRadius))

    def pcisph_init(self, particleRadius=0.025):
        self.lib.orc_pcisph_init(self.h, float(particleRadius))

    def iisph_init(self, particleRadius=0.025):
        self.lib.orc_iisph_init(self.h, float(particleRadius))

    def close(self):
        if self.__dict__.get("h"):
            self.lib.orc_destroy(self.h)
            self.__dict__["h"] = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
