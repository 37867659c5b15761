import math

import numpy as np


class CohesionKernel:
    """ref: kernels/CohesionKernel.py:12-33 (Akinci 2013 cohesion spline; `- m_c` outside the m_k factor, Q9)."""

    def __init__(self, searchR):
        self.searchR = searchR
        self.m_k = 32.0 / (math.pi * math.pow(searchR, 9.0))
        self.m_c = math.pow(searchR, 6.0) / 64.0

    def Cubic_W_norm(self, r):
        r = np.asarray(r, dtype=np.float32)
        h = np.float32(self.searchR)
        r3 = r * r * r
        far = np.float32(self.m_k) * (h - r) ** 3 * r3
        near = np.float32(self.m_k) * np.float32(2.0) * (h - r) ** 3 * r3 - np.float32(self.m_c)
        return np.where(r * r <= h * h, np.where(r > 0.5 * h, far, near), np.float32(0.0)).astype(np.float32)

    def Cubic_W(self, r_vec):
        return self.Cubic_W_norm(np.linalg.norm(np.asarray(r_vec, np.float32), axis=-1))
