from ._akinci import F, RadialSpline, cohesion_constants, np


class CohesionKernel(RadialSpline):
    """ref: kernels/CohesionKernel.py:12-33.  Outer half of the support: m_k (h-r)^3 r^3; inner half: twice that minus
    m_c -- with m_c OUTSIDE the m_k factor, as the reference has it (Q9)."""

    def __init__(self, searchR):
        super().__init__(searchR)
        self.m_k, self.m_c = cohesion_constants(searchR)

    def profile(self, dist, h):
        lobe = F(self.m_k) * (h - dist) ** 3 * (dist * dist * dist)
        return np.where(dist > F(0.5) * h, lobe, F(2.0) * lobe - F(self.m_c))
