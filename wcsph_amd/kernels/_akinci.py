"""Host-side (NumPy, float32) evaluation of the two surface-tension splines of Akinci et al. 2013 that the reference
keeps in kernels/CohesionKernel.py and kernels/AdhesionKernel.py.  The device evaluates its own copies
(csrc/dfsph_kernels.h); these classes exist for the reference's Python surface and for the host-logic tests."""
import math

import numpy as np

F = np.float32


class RadialSpline:
    """W(|r|) with compact support `searchR`; subclasses supply `profile(dist, h)` for 0 <= dist <= h."""

    def __init__(self, searchR):
        self.searchR = searchR

    def profile(self, dist, h):   # pragma: no cover - abstract
        raise NotImplementedError

    def Cubic_W_norm(self, r):
        dist = np.asarray(r, dtype=F)
        h = F(self.searchR)
        with np.errstate(invalid="ignore"):
            inside = self.profile(dist, h)
        return np.where(dist * dist <= h * h, inside, F(0.0)).astype(F)

    def Cubic_W(self, r_vec):
        return self.Cubic_W_norm(np.linalg.norm(np.asarray(r_vec, F), axis=-1))


def cohesion_constants(h):
    """(m_k, m_c) of the cohesion spline, ref: kernels/CohesionKernel.py:15-16"""
    return 32.0 / (math.pi * math.pow(h, 9.0)), math.pow(h, 6.0) / 64.0


def adhesion_constant(h):
    """m_k of the adhesion spline, ref: kernels/AdhesionKernel.py:15"""
    return 0.007 / math.pow(h, 3.25)
