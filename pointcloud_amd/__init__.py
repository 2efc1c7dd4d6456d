"""MI355X-native drop-in for the slice -> circle fit -> boundary -> residual
path of the thenamvn/pointcloud cylinder analyzer.

The modules mirror the reference's layout and call signatures
(``analyzer.Config`` / ``CylinderAnalyzer``, ``utils.circle_fitting``,
``utils.boundary``) and hand numpy arrays through a ctypes C ABI
(include/pcfit.h) to hand-written HIP kernels.  There is no CPU fallback.
"""
from .analyzer import Config, CylinderAnalyzer  # noqa: F401
from ._ffi import DeviceScan  # noqa: F401

__all__ = ["Config", "CylinderAnalyzer", "DeviceScan"]
