"""gpar_amd — MI355X-native (gfx950) per-layer GP inference hot path of GPAR behind the reference's API.

    from gpar_amd import GPARRegressor        # drop-in for gpar.GPARRegressor (reference gpar/__init__.py:1-2)

Importing the package does not touch the GPU; the first numerical call creates the HIP engine (and raises if
libgpar_hip.so or the GPU is missing: there is no CPU fallback).
"""
__version__ = "0.2.0"

# (Nothing process-wide is changed by importing the package.  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware
# queues - 4 by default - and streams that share a queue serialise; `HipEngine` asks for 8 when it is created before the
# process has touched the GPU and otherwise works with what the runtime already has: see engine._hardware_queues.)

from .regression import GPARRegressor, log_transform, squishing_transform  # noqa: F401
from .model import GPAR  # noqa: F401

__all__ = ["GPARRegressor", "GPAR", "log_transform", "squishing_transform"]
