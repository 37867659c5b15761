import math

import numpy as np


class AdhesionKernel:
    """ref: kernels/AdhesionKernel.py:12-33 (Akinci 2013 adhesion spline)."""

    def __init__(self, searchR):
        self.searchR = searchR
        self.m_k = 0.007 / math.pow(searchR, 3.25)

    def Cubic_W_norm(self, r):
        r = np.asarray(r, dtype=np.float32)
        h = np.float32(self.searchR)
        with np.errstate(invalid="ignore"):
            v = np.float32(self.m_k) * np.power(np.maximum(-4.0 * r * r / h + 6.0 * r - 2.0 * h, 0.0), 0.25)
        return np.where((r * r <= h * h) & (r > 0.5 * h), v, np.float32(0.0)).astype(np.float32)

    def Cubic_W(self, r_vec):
        return self.Cubic_W_norm(np.linalg.norm(np.asarray(r_vec, np.float32), axis=-1))
