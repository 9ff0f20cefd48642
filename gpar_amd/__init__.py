"""gpar_amd — MI355X-native (gfx950) per-layer GP inference hot path of GPAR behind the reference's API.

    from gpar_amd import GPARRegressor        # drop-in for gpar.GPARRegressor (reference gpar/__init__.py:1-2)

Importing the package does not touch the GPU; the first numerical call creates the HIP engine (and raises if
libgpar_hip.so or the GPU is missing: there is no CPU fallback).
"""
__version__ = "0.2.0"

import os as _os

# The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and streams that share a queue
# serialise.  Independent layers run on up to four streams beside the caller's, each factorisation may add a look-ahead side
# stream: with four queues the fourth layer stream silently ran behind the first (C2: 5.4 ms, 4.85 with eight queues).  Read
# by the runtime when it initialises, i.e. at the first GPU call of the process - a value set by the user wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .regression import GPARRegressor, log_transform, squishing_transform  # noqa: E402,F401
from .model import GPAR  # noqa: E402,F401

__all__ = ["GPARRegressor", "GPAR", "log_transform", "squishing_transform"]
