import math

import numpy as np


class CubicKernel:
    """ref: kernels/CubicKernel.py:12-54.  W(r) = m_k h3 P(q), gradW = m_l h3 {q(3q-2) | -(1-q)^2} r/(|r| h)."""

    def __init__(self, searchR):
        self.searchR = searchR
        self.h3 = 1.0 / (searchR * searchR * searchR)
        self.m_k = 8.0 / math.pi
        self.m_l = 48.0 / math.pi

    def Cubic_W_P(self, q):
        q = np.asarray(q, dtype=np.float32)
        f = np.float32(1.0) - q
        inner = np.float32(6.0) * q * q * q - np.float32(6.0) * q * q + np.float32(1.0)
        outer = np.float32(2.0) * f * f * f
        return np.where(q <= 1.0, np.where(q <= 0.5, inner, outer), np.float32(0.0)).astype(np.float32)

    def Cubic_W_norm(self, r):
        return (self.Cubic_W_P(np.asarray(r, np.float32) / np.float32(self.searchR)) * np.float32(self.m_k * self.h3)).astype(np.float32)

    def Cubic_W(self, r_vec):
        return self.Cubic_W_norm(np.linalg.norm(np.asarray(r_vec, np.float32), axis=-1))

    def CubicGradW(self, r_vec):
        r = np.asarray(r_vec, dtype=np.float32)
        rl = np.linalg.norm(r, axis=-1, keepdims=True).astype(np.float32)
        q = rl / np.float32(self.searchR)
        with np.errstate(divide="ignore", invalid="ignore"):
            gradq = r / (rl * np.float32(self.searchR))
            f = np.float32(1.0) - q
            s = np.where(q <= 0.5, q * (np.float32(3.0) * q - np.float32(2.0)), -(f * f)) * np.float32(self.m_l * self.h3)
            out = np.where((rl > 1.0e-5) & (q <= 1.0), s * gradq, np.float32(0.0))
        return out.astype(np.float32)
