"""gpar_amd — MI355X-native (gfx950) per-layer GP inference hot path of GPAR behind the reference's API.

    from gpar_amd import GPARRegressor        # drop-in for gpar.GPARRegressor (reference gpar/__init__.py:1-2)

Importing the package does not touch the GPU; the first numerical call creates the HIP engine (and raises if
libgpar_hip.so or the GPU is missing: there is no CPU fallback).
"""
__version__ = "0.1.0"

from .regression import GPARRegressor, log_transform, squishing_transform  # noqa: E402,F401
from .model import GPAR  # noqa: E402,F401

__all__ = ["GPARRegressor", "GPAR", "log_transform", "squishing_transform"]
