"""Every environment switch DESIGN.md section 3.9 documents is part of the behaviour surface: each one, set to a non-default
value, must leave a log marginal likelihood unchanged to 1e-11 (they select launch shapes, fusion and concurrency, never
arithmetic that matters).  One n = 1300 problem (ragged: not a multiple of any tile size) plus one n = 5200 factorisation for
the switches that only act on large matrices.  Switches read at call time are flipped in-process; the three that the library
caches on first use are exercised in a fresh process."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from .conftest import make_engine
from .test_parity_gpu import _problem

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CALL_TIME = [
    ("GPAR_POTRF_NBO", "256"),
    ("GPAR_PANEL_PROGRESSIVE", "0"),
    ("GPAR_PANEL_SPLIT", "0"),
    ("GPAR_PANEL_SPLIT", "2"),
    ("GPAR_PANEL_TILE_ROWS", "0"),
    ("GPAR_SPIN_CHAIN", "0"),
    ("GPAR_POTRF_FUSED", "0"),
    ("GPAR_POTRF_LOOKAHEAD", "0"),
    ("GPAR_POTRF_LOOKAHEAD", "1"),
    ("GPAR_POTRF_GROUP", "2"),
    ("GPAR_POTRF_GROUP", "1"),
    ("GPAR_POTRF_PAIR_ROWS", "1024"),
    ("GPAR_INVERSE_RECURSIVE", "0"),
    ("GPAR_TRSM_FUSED", "0"),
    ("GPAR_TRSV", "0"),
    ("GPAR_TRSM_GROUP", "2"),
    ("GPAR_TRSM_PAIRS", "0"),
    ("GPAR_PANEL_PAIRS", "0"),
    ("GPAR_HOST_MASKS", "0"),
    ("GPAR_TRSM_PAIR_COLS", "1024"),
    ("GPAR_LAYER_PIPELINE", "0"),
    ("GPAR_LAYER_PIPELINE", "3"),
    ("GPAR_LAYER_BATCH_ROWS", "0"),
    ("GPAR_LAYER_BATCH_BYTES", str(8 * 1301 * 1317 * 2)),
    ("GPAR_POTRF_BATCH_LOOKAHEAD", "0"),
    ("GPAR_POTRF_PREZERO", "0"),
    ("GPAR_POTRF_SMALL_UPDATE", "0"),
    ("GPAR_POTRF_LA_SMALL_TILES", "0"),
    ("GPAR_POTRF_LA_SMALL_TILES", "100000"),
    ("GPAR_POTRF_FUSE2_ROWS", "0"),
    ("GPAR_POTRF_FUSE2_ROWS", "100000"),
    ("GPAR_POTRF_FUSE2_BATCH_ROWS", "0"),
    ("GPAR_POTRF_FUSE2_BATCH_ROWS", "100000"),
    ("GPAR_POTRF_LA_SMALL_TILES2", "0"),
    ("GPAR_POTRF_LOCKSTEP_MIN", "2000"),
    ("GPAR_POTRF_FUSE_MAX", "2"),
    ("GPAR_POTRF_FUSE_MAX", "8"),
    ("GPAR_ONE_CALL", "0"),
    ("GPAR_LOCKSTEP_FUSED_BUILD_ROWS", "0"),
    ("GPAR_VFE_SPREAD_MAX", "1e3"),
    ("GPAR_LINEAR_TAIL", "0"),
    ("GPAR_GEMV", "0"),
    ("GPAR_VFE_FUSED_SCALARS", "0"),
    ("GPAR_ONE_CALL_GRAD_ROWS", "0"),
    ("GPAR_FIT_THREADS", "1"),
    ("GPAR_FIT_LOCKSTEP_ROWS", "1000"),   # (what it changes - concurrent fits - is tests/test_fastfit.py::test_lockstep_rendezvous_*)
    ("GPAR_NOTPD_RETRY", "0"),
    ("GPAR_GRAM_JIT_MIN_ENTRIES", "0"),
    ("GPAR_GRAM_JIT_MIN_ENTRIES", "-1"),
    ("GPAR_GRAD_JIT_MIN_ENTRIES", "0"),
    ("GPAR_GRAD_JIT_MIN_ENTRIES", "-1"),
]
CACHED = [("GPAR_AOT", "0"), ("GPAR_AOT_MIN_ENTRIES", "0"), ("GPAR_GEMM_HALF_TILES", "0"), ("GPAR_GEMM_HALF_TILES", "100000"), ("GPAR_GEMM_MIXED_TAIL", "0"),
          ("GPAR_GEMM_HALF_TILES_TRIANGULAR", "0")]


def _evaluate():
    """(logpdf of a 4-layer model at n = 1300, posterior logpdf of 700 points given 1000, gradient checksum, logdet at n = 5200)."""
    import torch

    from gpar_amd import hip as H
    from gpar_amd.engine import get_engine
    from gpar_amd.regression import GPARRegressor

    x, y = _problem(1300, 2, 4, seed=77)
    reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1, normalise_y=False)
    value = float(reg.logpdf(x, y))
    # a posterior quantity that involves every solve kernel and no sampling (samples of a numerically singular latent covariance
    # are only determined to sqrt(jitter)): the posterior log-density of held-out observations at 700 points
    reg.condition(x[:1000], y[:1000])
    mean = float(reg.logpdf(x[600:], y[600:], posterior=True))
    reg.vs.requires_grad(True)
    reg.logpdf(torch.tensor(x), torch.tensor(y)).backward()
    grad = float(sum(v.grad.abs().sum() for v in reg.vs.get_vars()))
    reg.vs.requires_grad(False)
    dev = get_engine().device
    g = torch.Generator().manual_seed(5)
    pts = torch.rand(5200, 3, generator=g, dtype=torch.float64).to(dev)
    A = H.alloc_matrix(5200, 5200, dev)
    A.copy_(torch.exp(-0.5 * torch.cdist(pts, pts) ** 2 / 0.25))
    A.diagonal().add_(0.1)
    logdet, info = H.potrf_(A)
    assert int(info.item()) == 0
    return value, mean, grad, float(logdet)


def _close(a, b):
    for u, v, tol in zip(a, b, (1e-11, 1e-9, 1e-8, 1e-11)):
        assert abs(u - v) <= tol * abs(v), (a, b)


@pytest.fixture(scope="module")
def baseline():
    from gpar_amd.engine import set_engine

    eng = make_engine("hip")
    previous = set_engine(eng)
    try:
        yield _evaluate()
    finally:
        set_engine(previous)


@pytest.mark.parametrize("name,value", CALL_TIME, ids=[f"{k}={v}" for k, v in CALL_TIME])
def test_call_time_switch_leaves_the_results_alone(baseline, monkeypatch, name, value):
    from gpar_amd.engine import set_engine

    monkeypatch.setenv(name, value)
    eng = make_engine("hip")
    previous = set_engine(eng)
    try:
        _close(_evaluate(), baseline)
    finally:
        set_engine(previous)


@pytest.mark.parametrize("name,value", CACHED, ids=[f"{k}={v}" for k, v in CACHED])
def test_cached_switch_leaves_the_results_alone(baseline, name, value):
    code = ("import json, sys; sys.path.insert(0, %r); from tests.test_switches_gpu import _evaluate; from tests.conftest import make_engine; "
            "from gpar_amd.engine import set_engine; set_engine(make_engine('hip')); print(json.dumps(_evaluate()))" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **{name: value}), cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    _close(json.loads(out.stdout.strip().splitlines()[-1]), baseline)


def test_every_documented_switch_is_exercised():
    """The switch table of DESIGN.md and this file list the same variables (experiment knobs of section 7 excepted)."""
    import re

    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    table = text[text.index("### 3.9 Switches"):text.index("## 4. Oracle and parity")]
    rows = [line for line in table.splitlines() if line.startswith("| `") and "experiment knobs" not in line]
    documented = set(re.findall(r"`(GPAR_[A-Z_]+)`", "\n".join(row.split("|")[1] for row in rows)))
    exercised = {k for k, _ in CALL_TIME + CACHED}
    assert documented <= exercised, documented - exercised
