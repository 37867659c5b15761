"""wcsph_amd -- MI355X-native SPH hot path behind the lyd405121/wcsph Python surface.

Public surface (same names as the reference modules):
    ParticleData, HashGrid                       (ref: ParticleData.py, HashGrid.py)
    kernels.CubicKernel / CohesionKernel / AdhesionKernel
    dfsph, sesph                                 (driver modules: same step-function names)
The compute path is libwcsph_hip.so (hand-written gfx950 kernels, include/wcsph.h) called
through ctypes.  There is NO CPU fallback: using any field or kernel without the built
library and a GPU raises.
"""
from .ParticleData import ParticleData  # noqa: F401
from .HashGrid import HashGrid  # noqa: F401
from . import kernels  # noqa: F401

__all__ = ["ParticleData", "HashGrid", "kernels"]
