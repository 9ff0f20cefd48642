"""The product against the restatement that shares NOTHING with it (`oracle/gpar_ref.py` + `oracle/gp_ref.py`: its own layer
kernels built from the hyper-parameter dictionary, its own missing-data bookkeeping and normalisation, dense `slogdet` + `solve`
instead of any factorisation the product composes), beyond the log marginal likelihood:

  * conditioning chain + predictive means / variances for `replace=True`, where they exist in closed form
    (reference gpar/model.py:116-149, 245-277, 291-322; gpar/regression.py:339-389, 566-597) - against the product's
    `predict_moments`, and the Monte-Carlo `predict` against both;
  * the posterior SAMPLES themselves (reference gpar/model.py:245-277: the noisy draw of a layer is the next layer's input) - in
    distribution, against `gpar_ref.gpar_sample` with its own random numbers: means, spreads and the correlation between consecutive
    outputs, with the linear-output shortcut, the general path, a Markov window and `replace`;
  * the gradient `fit` needs (reference gpar/regression.py:434-459; autograd there, analytic kernels here) - against central
    finite differences of `gpar_ref.gpar_logpdf` in the hyper-parameter dictionary, mapped to the optimiser's variables.

Which tests share host algebra with the product and which do not: tests that run the SAME product code on the numpy engine and on
the HIP engine (`tests/test_parity_gpu.py::test_logpdf_condition_predict_match_oracle`, the fuzz tests, `test_gradient_matches_oracle`)
test the kernels, not the formulas; the tests in THIS file (and `test_logpdf_matches_the_independent_oracle_route`,
`tests/test_reference_golden.py`, the golden vectors) test the formulas: every number on the right-hand side comes from code that
imports nothing from `gpar_amd`.

Every test takes the `engine` fixture: on the CPU (`-m "not gpu"`) the product's host algebra runs on the numpy engine, on the
MI355X (`-m gpu`) the same calls go through libgpar_hip.so.
"""
import numpy as np
import pytest
import torch

# (shape-alikes of the BASELINE configurations at sizes the dense restatement finishes in seconds; as tests/test_parity_gpu.py)
CONFIGS = {
    "C1-paper-synthetic": (dict(scale=0.1, linear=True, linear_scale=10.0, nonlinear=True, nonlinear_scale=0.1, noise=0.1, normalise_y=False), 25, 1, 3, 0.0),
    "C2-shape": (dict(scale=0.5, linear=True, nonlinear=False, noise=0.1), 384, 2, 4, 0.0),
    "C3-shape-markov2": (dict(scale=0.5, linear=True, nonlinear=True, markov=2, noise=0.1), 300, 4, 8, 0.0),
    "C4-shape-inducing": (dict(scale=0.5, linear=True, nonlinear=True, noise=0.1, x_ind=np.random.default_rng(8).uniform(0, 1, (64, 8))), 500, 8, 4, 0.0),
    "C5-shape-per-rq": (dict(scale=0.5, per=True, rq=True, linear=True, nonlinear=True, noise=0.1), 200, 3, 5, 0.0),
    "missing-data": (dict(scale=0.5, linear=True, nonlinear=True, noise=0.1), 257, 2, 3, 0.2),
}


def _problem(n, m, p, seed, missing=0.0):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 1, (n, m))
    cols = []
    for i in range(p):
        base = np.sin(2 * np.pi * (x @ rng.uniform(0.5, 1.5, m)) + i)
        if cols:
            base = base + 0.5 * cols[-1] ** 2
        cols.append(base + 0.1 * rng.standard_normal(n))
    y = np.stack(cols, axis=1)
    y = 0.7 + 1.9 * (y - y.mean(0)) / y.std(0)   # (not standardised: the normalisation of `condition` has something to do)
    if missing:
        y[rng.random(y.shape) < missing] = np.nan
        y[0] = 0.25
    return x, y


@pytest.mark.parametrize("latent", [False, True], ids=["observed", "latent"])
@pytest.mark.parametrize("name", list(CONFIGS))
def test_predict_moments_match_the_share_nothing_route(engine, name, latent):
    """condition + closed-form predict (replace=True) at every configuration shape: means and variances at 40 new inputs, with
    weights on the new points, rtol 1e-8 (inducing points: 1e-6 - both routes go through K_zz^-1, conditioned by its jitter)."""
    from gpar_amd.regression import GPARRegressor
    from oracle import gpar_ref

    kw, n, m, p, missing = CONFIGS[name]
    kw = dict(kw, replace=True, impute=True)
    x, y = _problem(n, m, p, seed=len(name) + 3, missing=missing)
    rng = np.random.default_rng(5)
    xs = rng.uniform(0, 1, (40, m))
    ws = rng.uniform(0.5, 2.0, (40, p))
    reg = GPARRegressor(**kw)
    reg.condition(x, y)
    mean, var = reg.predict_moments(xs, ws, latent=latent)
    want_mean, want_var = gpar_ref.gpar_predict_moments(x, y, None, reg.get_variables(), reg.model_config, xs, ws, latent=latent,
                                                        impute=True, replace=True, x_ind=kw.get("x_ind"),
                                                        normalise_y=kw.get("normalise_y", True))
    assert mean.shape == var.shape == (40, p)
    rtol = 1e-8 if kw.get("x_ind") is None else 1e-6
    np.testing.assert_allclose(mean, want_mean, rtol=rtol, atol=rtol * np.max(np.abs(want_mean)))
    np.testing.assert_allclose(var, want_var, rtol=rtol, atol=rtol * np.max(np.abs(want_var)))
    assert np.all(var > 0)


def test_monte_carlo_predict_converges_to_the_closed_form(engine):
    """`predict` (the reference's Monte-Carlo estimate, regression.py:566-597) with replace=True against the share-nothing closed
    form: the mean of S samples within 5 standard errors at every point, the sample spread within 25 % of the closed-form
    standard deviation, the central 95 % bounds around mean -+ 1.96 sd."""
    from gpar_amd.regression import GPARRegressor
    from oracle import gpar_ref

    x, y = _problem(150, 2, 3, seed=11)
    xs = np.random.default_rng(2).uniform(0, 1, (25, 2))
    reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1, replace=True, impute=True)
    reg.condition(x, y)
    S = 600
    mean, lower, upper = reg.predict(xs, num_samples=S, credible_bounds=True)
    want_mean, want_var = gpar_ref.gpar_predict_moments(x, y, None, reg.get_variables(), reg.model_config, xs, impute=True, replace=True)
    sd = np.sqrt(want_var)
    assert np.all(np.abs(mean - want_mean) <= 5.0 * sd / np.sqrt(S))
    np.testing.assert_allclose(upper - lower, 2 * 1.96 * sd, rtol=0.25)
    np.testing.assert_allclose(0.5 * (upper + lower), want_mean, atol=0.35 * np.max(sd))


def test_predict_moments_refuses_what_has_no_closed_form(engine):
    from gpar_amd.regression import GPARRegressor

    x, y = _problem(40, 1, 2, seed=1)
    reg = GPARRegressor(replace=False)
    with pytest.raises(RuntimeError):
        reg.predict_moments(x)
    reg.condition(x, y)
    with pytest.raises(ValueError):
        reg.predict_moments(x)
    reg = GPARRegressor(replace=True, transform_y=(torch.log, torch.exp))
    reg.condition(x, np.abs(y) + 0.1)
    with pytest.raises(ValueError):
        reg.predict_moments(x)


GRADIENT_CONFIGS = {
    # independent layers (data as inputs), the chain through imputed means, through replaced means, and through inducing inputs
    "C2-shape": (dict(scale=0.5, linear=True, nonlinear=False, noise=0.1), 120, 2, 4, 0.0),
    "C3-shape-markov2": (dict(scale=0.5, linear=True, nonlinear=True, markov=2, noise=0.1), 90, 4, 8, 0.0),
    "C4-shape-inducing": (dict(scale=0.5, linear=True, nonlinear=True, noise=0.1, x_ind=np.random.default_rng(8).uniform(0, 1, (24, 3))), 110, 3, 4, 0.0),
    "C5-shape-per-rq": (dict(scale=0.5, per=True, rq=True, input_linear=True, linear=True, nonlinear=True, noise=0.1), 80, 3, 4, 0.0),
    "missing-impute": (dict(scale=0.5, linear=True, nonlinear=True, noise=0.1, impute=True), 100, 2, 3, 0.2),
    "impute+replace": (dict(scale=0.5, linear=True, nonlinear=True, rq=True, noise=0.1, impute=True, replace=True), 100, 2, 3, 0.15),
}


@pytest.mark.parametrize("name", list(GRADIENT_CONFIGS))
def test_joint_gradient_matches_share_nothing_finite_differences(engine, name):
    """d logpdf / d(every optimiser variable) - what `fit` hands L-BFGS-B, analytic here (`gpar_chol_inverse`, `gpar_gram_grad`,
    the exact joint gradient through forwarded means) - against central differences of the share-nothing `gpar_ref.gpar_logpdf`
    in every entry of the hyper-parameter dictionary, carried to the optimiser's variables by the bounded map restated in
    `gpar_ref.to_unconstrained`: 1e-6 of the largest component."""
    from gpar_amd.regression import GPARRegressor
    from oracle import gpar_ref

    kw, n, m, p, missing = GRADIENT_CONFIGS[name]
    x, y = _problem(n, m, p, seed=len(name) + 40, missing=missing)
    y = (y - np.nanmean(y, axis=0)) / np.nanstd(y, axis=0)
    reg = GPARRegressor(**dict(kw, normalise_y=False))
    with torch.no_grad():
        reg.logpdf(x, y)   # (the variables exist from the first evaluation on)
    reg.vs.requires_grad(True)
    reg.logpdf(torch.tensor(x), torch.tensor(y)).backward()
    hypers = reg.get_variables()
    fd = gpar_ref.fd_gradient(x, y, None, hypers, reg.model_config, impute=reg.impute, replace=reg.replace, x_ind=kw.get("x_ind"))
    got, want = [], []
    for var_name in reg.vs.names:
        latent = reg.vs.get_vars(var_name)[0]
        g = latent.grad if latent.grad is not None else torch.zeros_like(latent)
        got.append(g.detach().cpu().numpy().reshape(-1))
        want.append(gpar_ref.to_unconstrained(var_name, hypers[var_name], fd[var_name]).reshape(-1))
    got, want = np.concatenate(got), np.concatenate(want)
    assert np.max(np.abs(want)) > 1e-2
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6 * np.max(np.abs(want)))


SAMPLE_CONFIGS = {
    # the reference's default output dependence (linear only: the product takes its shared-solve shortcut, gp.Obs._sample_batch_linear_tail),
    # the general path (stacked solves, batched downdates), a Markov window, the chain through replaced means
    "linear-outputs": (dict(scale=0.5, linear=True, nonlinear=False, noise=0.05), dict()),
    "nonlinear-outputs": (dict(scale=0.5, linear=True, nonlinear=True, noise=0.05), dict()),
    "markov1-rq": (dict(scale=0.5, linear=True, nonlinear=True, rq=True, markov=1, noise=0.05), dict()),
    "replace": (dict(scale=0.5, linear=True, nonlinear=True, noise=0.05, replace=True, impute=True), dict()),
}


@pytest.mark.parametrize("latent", [False, True], ids=["observed", "latent"])
@pytest.mark.parametrize("name", list(SAMPLE_CONFIGS))
def test_posterior_samples_have_the_share_nothing_samplers_distribution(engine, name, latent):
    """`sample(posterior=True)` - joint ancestral sampling, the noisy draw of a layer fed to the next (reference model.py:245-277) -
    against `gpar_ref.gpar_sample`, a numpy restatement with its own random numbers: per test point and output the means of S = 1500
    draws agree within 5 standard errors, the standard deviations within 15 %, and so does the correlation between consecutive
    outputs at a point (within 0.16: four standard errors of a difference of two sample correlations) - what feeding SAMPLES (not means) forward produces.  No shared code, no shared randomness."""
    from gpar_amd.regression import GPARRegressor
    from oracle import gpar_ref

    kw, _ = SAMPLE_CONFIGS[name]
    x, y = _problem(70, 2, 3, seed=len(name) + 71)
    xs = np.random.default_rng(3).uniform(0, 1, (6, 2))
    ws = np.random.default_rng(4).uniform(0.5, 2.0, (6, 3))
    reg = GPARRegressor(**kw)
    reg.condition(x, y)
    S = 1500
    got = np.stack(reg.sample(xs, ws, posterior=True, num_samples=S, latent=latent))
    want = gpar_ref.gpar_sample(x, y, None, reg.get_variables(), reg.model_config, xs, ws, num_samples=S, latent=latent,
                                impute=kw.get("impute", False), replace=kw.get("replace", False), seed=9)
    assert got.shape == want.shape == (S, 6, 3)
    mg, mw, sg, sw = got.mean(0), want.mean(0), got.std(0), want.std(0)
    assert np.all(np.abs(mg - mw) <= 5.0 * np.sqrt((sg ** 2 + sw ** 2) / S)), np.max(np.abs(mg - mw) / np.sqrt((sg ** 2 + sw ** 2) / S))
    np.testing.assert_allclose(sg, sw, rtol=0.15)
    for i in range(2):
        cg = [np.corrcoef(got[:, j, i], got[:, j, i + 1])[0, 1] for j in range(6)]
        cw = [np.corrcoef(want[:, j, i], want[:, j, i + 1])[0, 1] for j in range(6)]
        np.testing.assert_allclose(cg, cw, atol=0.16)
