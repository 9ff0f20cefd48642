import os
import sys

import numpy as np
import pytest

# the product asks for 8 hardware queues when its engine is created before the GPU is touched; the test session touches the GPU
# earlier (torch.cuda.is_available() below), so the request is made here, as an application would export it
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:  # pragma: no cover
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests are only meaningful where a GPU is visible; elsewhere they are skipped, not failed.
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def make_engine(kind, seed=1234):
    """'oracle': the numpy CPU oracle (test infrastructure); 'hip': the product engine on cuda:0."""
    if kind == "oracle":
        from oracle.engine import OracleEngine

        return OracleEngine(seed=seed)
    from gpar_amd.engine import HipEngine

    return HipEngine(seed=seed)


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def engine(request):
    """Installs the engine the host code runs on.  `-m "not gpu"` exercises the host logic on the CPU oracle;
    `-m gpu` runs the very same behavioural tests through libgpar_hip.so."""
    from gpar_amd.engine import set_engine

    eng = make_engine(request.param)
    previous = set_engine(eng)
    yield eng
    set_engine(previous)


@pytest.fixture
def oracle_engine():
    from gpar_amd.engine import set_engine

    eng = make_engine("oracle")
    previous = set_engine(eng)
    yield eng
    set_engine(previous)


def to_np(a):
    """numpy view of numpy / torch (any device) / tuples thereof."""
    import torch

    if isinstance(a, torch.Tensor):
        return a.detach().cpu().numpy()
    return np.asarray(a)


def close(a, b, rtol=1e-7, atol=1e-12):
    """assert_allclose with the reference test-suite's default tolerances (tests/util.py:11-20 there), recursing
    into tuples."""
    if isinstance(a, tuple) or isinstance(b, tuple):
        assert isinstance(a, tuple) and isinstance(b, tuple) and len(a) == len(b)
        for u, v in zip(a, b):
            close(u, v, rtol=rtol, atol=atol)
        return
    np.testing.assert_allclose(to_np(a), to_np(b), rtol=rtol, atol=atol)


def columns_all_different(a, b, tol=1e-2):
    """Every column of a is further than `tol` from every column of b (reference tests/util.py:32-39)."""
    a, b = to_np(a), to_np(b)
    a = a.reshape(a.shape[0], -1) if a.ndim > 1 else a.reshape(-1, 1)
    b = b.reshape(b.shape[0], -1) if b.ndim > 1 else b.reshape(-1, 1)
    d = np.sqrt(((a.T[:, None, :] - b.T[None, :, :]) ** 2).sum(-1))
    assert np.all(d > tol)
