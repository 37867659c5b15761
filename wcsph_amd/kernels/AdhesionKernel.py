from ._akinci import F, RadialSpline, adhesion_constant, np


class AdhesionKernel(RadialSpline):
    """ref: kernels/AdhesionKernel.py:12-33.  Non-zero only on the outer half of the support:
    m_k (-4 r^2 / h + 6 r - 2 h)^(1/4); the radicand is clamped at 0 (it is -1e-9 at r = h in float32)."""

    def __init__(self, searchR):
        super().__init__(searchR)
        self.m_k = adhesion_constant(searchR)

    def profile(self, dist, h):
        radicand = np.maximum(F(-4.0) * dist * dist / h + F(6.0) * dist - F(2.0) * h, F(0.0))
        return np.where(dist > F(0.5) * h, F(self.m_k) * np.power(radicand, F(0.25)), F(0.0))
