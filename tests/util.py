"""Shared helpers of the parity tests: one small scene instantiated on both sides."""
import numpy as np

from oracle import Oracle
from wcsph_amd import ParticleData, scene
from wcsph_amd.field import Field, ScalarField


class GpuSim:
    """Thin handle on a wcsph_amd.ParticleData + the dfsph module-level fields, independent of module globals."""

    def __init__(self, liquid, solid, particleR=0.025, mode="faithful", maxInGrid=64, maxNeighbour=2048, **attrs):
        from wcsph_amd import _lib
        self.lib = _lib.load()
        self._lib = _lib
        pd = ParticleData(particleR, neighbor_mode=mode, maxInGrid=maxInGrid, maxNeighbour=maxNeighbour,
                          cfl_radius=attrs.pop("cfl_radius", None))
        for k, v in attrs.items():
            setattr(pd, k, v)
        pd.add_liquid_points(liquid)
        pd.add_solid_points(solid)
        pd.setup_data_gpu()
        pd.setup_data_cpu()
        self.pd = pd
        self.deltaT = ScalarField(pd, "deltaT")
        self.alpha_coff = Field(pd, "alpha_coff", 1)
        self.kappa = Field(pd, "kappa", 1)
        self.kappa_v = Field(pd, "kappa_v", 1)

    def call(self, name):
        ctx = self.pd._need_ctx()
        self._lib.check(ctx, getattr(self.lib, "wcsph_" + name)(ctx))

    def iters(self):
        import ctypes as C
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self.lib.wcsph_get_iters(self.pd._need_ctx(), C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def scalar(self, name):
        import ctypes as C
        v = C.c_float()
        ctx = self.pd._need_ctx()
        self._lib.check(ctx, self.lib.wcsph_get_scalar(ctx, self._lib.SCALAR_ID[name], C.byref(v)))
        return v.value

    def close(self):
        self.pd.close()


def small_scene(n=8, jitter=0.0, seed=20240901):
    """n^3 dam-break block + box shell (SURVEY 8(d) recipe), optional uniform jitter of +-jitter*R."""
    liquid, solid = scene.dam_break_dfsph(n)
    if jitter:
        rng = np.random.default_rng(seed)
        liquid = liquid + rng.uniform(-jitter, jitter, liquid.shape) * scene.PARTICLE_RADIUS
    return liquid, solid


def make_pair(n=8, mode="faithful", jitter=0.0, seed=20240901, **kw):
    liquid, solid = small_scene(n, jitter, seed)
    g = GpuSim(liquid, solid, mode=mode, **kw)
    pos0 = g.pd.pos.to_numpy()
    o = Oracle(pos0, len(liquid), 0.025, mode=mode, maxInGrid=kw.get("maxInGrid", 64),
               maxNeighbour=kw.get("maxNeighbour", 2048))
    return g, o


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    return np.abs(a - b).max() / scale


def set_state(g, o, **fields):
    """Write the same arrays into both simulations (original particle order)."""
    for name, val in fields.items():
        val = np.asarray(val, np.float32)
        if name == "deltaT":
            g.deltaT[0] = float(val)
            o.deltaT = float(val)
            continue
        if name in ("alpha_coff", "kappa", "kappa_v"):
            getattr(g, name).from_numpy(val)
        else:
            getattr(g.pd, name).from_numpy(val)
        getattr(o, name)[...] = val
