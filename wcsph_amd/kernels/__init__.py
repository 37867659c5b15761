"""SPH smoothing kernels with the reference constructor signatures and attribute names
(ref: kernels/CubicKernel.py, kernels/CohesionKernel.py, kernels/AdhesionKernel.py).
The device-side evaluation lives in csrc/device.h; these host classes carry the constants
and offer a NumPy evaluation for scene set-up and inspection."""
from .CubicKernel import CubicKernel  # noqa: F401
from .CohesionKernel import CohesionKernel  # noqa: F401
from .AdhesionKernel import AdhesionKernel  # noqa: F401
