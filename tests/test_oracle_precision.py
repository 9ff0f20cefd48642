"""How well the fp64 oracle is pinned (stheno itself cannot run here, so parity stays
"unpinned" against it - these tests bound everything else):

  * every committed golden value against 50-digit arithmetic (oracle/mp_ref.py): the fp64 closed-form routes that
    generated them lose at most a few 1e-13 relative;
  * the ONE documented numerical departure from the reference's stack - squared distances from explicit differences here,
    |a|^2 + |b|^2 - 2 a b^T in lab - quantified on log marginal likelihoods at benchmark
    conditioning: the restated reference path on torch-CPU operators (oracle/torch_cpu.py) against the oracle engine.
"""
import json

import numpy as np
import pytest

from oracle import gpar_ref, mp_ref

from .test_oracle import GOLDEN, _nan_array


def _golden():
    with open(GOLDEN) as f:
        return json.load(f)


@pytest.mark.parametrize("case", [c for c in _golden()["gpar_logpdf"] if len(c["x"]) <= 30 and c.get("x_ind") is None], ids=lambda c: c["name"])
def test_golden_gpar_logpdf_against_50_digits(case):
    x, y = np.array(case["x"]), _nan_array(case["y"])
    w = None if case["w"] is None else np.array(case["w"])
    if "train_y" in case:   # micro-normalise-quirk: the argument of logpdf goes through the un-normalising map (gpar/regression.py:483)
        ty = _nan_array(case["train_y"])
        y = y * np.array([np.std(c[~np.isnan(c)]) for c in ty.T]) + np.array([np.mean(c[~np.isnan(c)]) for c in ty.T])
    exact = mp_ref.gpar_logpdf(x, y, w, case["hypers"], case["config"], impute=case["impute"], replace=case["replace"],
                               eps=case.get("epsilon", 1e-12))
    assert abs(float((exact - case["logpdf"]) / exact)) < 5e-13, (float(exact), case["logpdf"])


@pytest.mark.parametrize("case", _golden()["single_gp"], ids=lambda c: c["name"])
def test_golden_posterior_moments_against_50_digits(case):
    spec, _ = gpar_ref.layer_spec(case["hypers"], 2, 0, case["config"])
    x, y, noise, xs = (np.array(case[k]) for k in ("x", "y", "noise", "xs"))
    rows = lambda a: [[float(v) for v in r] for r in a]
    exact = mp_ref.logpdf(spec, rows(x), y, noise)
    assert abs(float((exact - case["logpdf"]) / exact)) < 5e-13
    mean, cov = mp_ref.posterior(spec, rows(x), y, noise, rows(xs))
    mean = np.array([float(v) for v in mean])
    cov = np.array([[float(cov[i, j]) for j in range(cov.cols)] for i in range(cov.rows)])
    scale = np.max(np.abs(cov))
    np.testing.assert_allclose(case["mean"], mean, rtol=0, atol=1e-12 * max(1.0, np.max(np.abs(mean))))
    np.testing.assert_allclose(case["cov"], cov, rtol=0, atol=1e-12 * scale)


@pytest.mark.parametrize("case", _golden()["vfe"], ids=lambda c: c["name"])
def test_golden_inducing_point_bound_against_50_digits(case):
    spec, _ = gpar_ref.layer_spec(case["hypers"], 1, 0, case["config"])
    x, y, noise, z = (np.array(case[k]) for k in ("x", "y", "noise", "z"))
    rows = lambda a: [[float(v) for v in r] for r in a]
    exact = mp_ref.vfe_bound(spec, rows(x), y, noise, rows(z))
    # K_zz + 1e-12 I is ill-conditioned for 12 inducing points on [-2, 2]: the fp64 routes keep ~1e-9
    assert abs(float((exact - case["bound"]) / exact)) < 5e-9, (float(exact), case["bound"])


@pytest.mark.parametrize("n,m,p_cols,noise", [(2048, 2, [2, 3], 0.1), (4096, 4, [9, 10], 0.1), (2048, 3, [3], 1e-2)])
def test_expanded_versus_explicit_squared_distances(oracle_engine, n, m, p_cols, noise):
    """|a|^2 + |b|^2 - 2 a b^T (lab, the reference) loses ~eps |a|^2 absolutely in every squared distance; explicit
    differences (this repository, HIP and oracle alike) do not.  On the log marginal likelihood of benchmark-shaped layers
    (C2 / C3 kernels, unit-box inputs, standardised outputs) the two agree to ~1e-11 relative - the tolerance DESIGN.md
    section 4 states for parity with the reference stack; this test would catch a regression to 1e-9."""
    import torch

    from gpar_amd.gp import GP
    from gpar_amd.kernels import EQ, Linear
    from oracle import kernels as ok
    from oracle import torch_cpu as tc

    rng = np.random.default_rng(n + m)
    width = max(p_cols) + 1
    x = rng.uniform(0, 1, (n, width))
    x[:, m:] = rng.standard_normal((n, width - m))
    y = np.sin(3 * x[:, 0]) + 0.3 * x[:, -1] + 0.1 * rng.standard_normal(n)
    kernel = (1.0 * EQ().stretch(np.full(m, 0.5))).select(list(range(m))) + (
        Linear().stretch(np.full(len(p_cols), 100.0)) + 1.0 * EQ().stretch(np.ones(len(p_cols)))
    ).select(p_cols)
    explicit = float(GP(kernel)(x, noise).logpdf(y))
    spec = ok.spec_to_dict(kernel.resolve(width))
    tc.set_threads()
    expanded, _, _ = tc.layer_logpdf(spec, torch.as_tensor(x), y, np.full(n, noise))
    rel = abs(expanded - explicit) / abs(explicit)
    assert rel < 1e-10, (explicit, expanded, rel)
