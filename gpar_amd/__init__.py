"""gpar_amd — MI355X-native (gfx950) per-layer GP inference hot path of GPAR behind the reference's API."""
__version__ = "0.1.0"
