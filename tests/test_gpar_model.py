"""Behavioural spec of gpar_amd.model, restating what /root/reference/tests/test_model.py pins for gpar/model.py.

Every check is either an exact known answer of the reference's host logic (merge / last / per_output tables) or a
closed-form GP identity (dense logpdf, VFE bound with Z = X, noise-free interpolation, sum-of-layers
decomposition).  Each test runs on the CPU oracle engine (`-m "not gpu"`) and, unchanged, through the HIP library
(`-m gpu`).
"""
import numpy as np
import pytest
import torch

from gpar_amd.gp import GP, Measure, Obs, SparseObs
from gpar_amd.kernels import EQ, Linear
from gpar_amd.model import GPAR, construct_model, last, merge, per_output

from .conftest import close, columns_all_different, to_np

NAN = np.nan


@pytest.fixture(params=[1, 2])
def x(request):
    return np.random.default_rng(10 + request.param).standard_normal((10, request.param))


@pytest.fixture
def w():
    return np.random.default_rng(3).random((10, 2)) + 1e-2


# ---- host index logic: exact known answers (reference tests/test_model.py:30-105) -------------------------

def test_merge_known_answers():
    base, new = np.array([1, 2, 3, 4]), np.array([5, 6])
    assert merge(base, new, np.array([True, True, False, False])).tolist() == [5, 6, 3, 4]
    assert merge(base, new, np.array([True, False, True, False])).tolist() == [5, 2, 6, 4]
    t = merge(torch.tensor([1.0, 2, 3, 4]), torch.tensor([5.0, 6]), torch.tensor([False, True, False, True]))
    assert t.tolist() == [1, 5, 3, 6]
    assert base.tolist() == [1, 2, 3, 4]  # not modified in place


def test_construct_model_returns_its_arguments():
    assert construct_model(1, 2)() == (1, 2)


def test_last_known_answers():
    seq = [1, 2, 3, 4]
    assert list(last(seq)) == [(False, 1), (False, 2), (False, 3), (True, 4)]
    assert list(last(seq, [1, 2])) == [(False, 2), (False, 3)]
    assert list(last(seq, [0, 3])) == [(False, 1), (True, 4)]
    assert list(last([])) == [] and list(last([], [0, 1])) == []
    assert list(last(iter(seq), [3])) == [(True, 4)]  # generators work too (zip objects in GPAR.logpdf)


PATTERN = np.array(
    [
        [1, 2, NAN, NAN],
        [3, NAN, 4, NAN],
        [5, 6, 7, NAN],
        [8, NAN, NAN, NAN],
        [9, 10, NAN, NAN],
        [11, NAN, NAN, 12],
    ]
)
EXPECT_DROP = [
    ([1, 3, 5, 8, 9, 11], [True] * 6),
    ([2, 6, 10], [True, False, True, False, True, False]),
    ([7], [False, True, False]),
    ([], [False]),
]
EXPECT_KEEP = [
    ([1, 3, 5, 8, 9, 11], [True] * 6),
    ([2, None, 6, 10, None], [True, True, True, False, True, True]),
    ([4, 7, None], [False, True, True, False, True]),
    ([12], [False, False, True]),
]


@pytest.mark.parametrize("as_torch", [False, True])
@pytest.mark.parametrize("which", ["y", "w"])
def test_per_output_known_answers(as_torch, which):
    data = torch.tensor(PATTERN) if as_torch else PATTERN

    def run(keep):
        out = []
        for yi, wi, mask in per_output(data, data, keep=keep):
            assert to_np(yi).ndim == 2 and to_np(wi).ndim == 1
            vals = to_np(yi)[:, 0] if which == "y" else to_np(wi)
            out.append(([None if np.isnan(v) else v for v in vals.tolist()], to_np(mask).tolist()))
        return out

    assert run(False) == EXPECT_DROP
    assert run(True) == EXPECT_KEEP


def test_per_output_accepts_precomputed_cache():
    cache = {True: [2, 3], False: [4]}
    assert list(per_output(cache, None, keep=True)) == [2, 3]
    assert list(per_output(cache, None, keep=False)) == [4]


def test_sparse_flag():
    assert not GPAR(x_ind=None).sparse and GPAR(x_ind=None).x_ind is None
    g = GPAR(x_ind=1)
    assert g.sparse and g.x_ind == 1


# ---- GP identities (reference tests/test_model.py:118-293) -----------------------------------------------

def test_observations_drop_missing_and_weight_noise(engine, x):
    prior = Measure()
    f = GP(EQ(), measure=prior)
    noise = 0.1
    w1 = np.random.default_rng(0).random(x.shape[0]) + 1e-2
    y = f(x, 0.1).sample()
    y_missing = y.clone()
    y_missing[::2] = NAN
    direct = f(x[1::2], noise / w1[1::2]).logpdf(y[1::2])

    obs = GPAR()._obs(x, None, y_missing, w1, f, noise)
    assert isinstance(obs, Obs)
    close(prior.logpdf(obs), direct, atol=1e-6)

    # inducing points at the data: the VFE bound is tight
    obs = GPAR(x_ind=x)._obs(x, x, y_missing, w1, f, noise)
    assert isinstance(obs, SparseObs)
    close(prior.logpdf(obs), direct, atol=1e-6)


def test_update_inputs_prior_and_posterior(engine):
    f = GP(EQ(), measure=Measure())
    x = np.array([[1.0], [2], [3]])
    y = np.array([[4.0], [5], [6]])
    xi = np.array([[6.0], [7]])
    y_hole = y.copy()
    y_hole[1] = NAN

    def expect(col, ind_col):
        return np.concatenate([x, np.array(col, dtype=float)[:, None]], 1), np.concatenate([xi, np.array(ind_col, dtype=float)[:, None]], 1)

    # no observations: the estimate is the prior mean, zero
    close(GPAR(x_ind=xi)._update_inputs(x, xi, y, f, None), expect([4, 5, 6], [0, 0]))
    close(GPAR(impute=True, x_ind=xi)._update_inputs(x, xi, y_hole, f, None), expect([4, 0, 6], [0, 0]))
    close(GPAR(replace=True, x_ind=xi)._update_inputs(x, xi, y_hole, f, None), expect([0, NAN, 0], [0, 0]))
    close(GPAR(impute=True, replace=True, x_ind=xi)._update_inputs(x, xi, y, f, None), expect([0, 0, 0], [0, 0]))

    # noise-free observations 9..13 at 1,2,3,6,7: the posterior mean interpolates them
    obs = Obs(f(np.array([1.0, 2, 3, 6, 7])), np.array([9.0, 10, 11, 12, 13]))
    close(GPAR(impute=True, x_ind=xi)._update_inputs(x, xi, y_hole, f, obs), expect([4, 10, 6], [12, 13]), atol=1e-6)
    close(GPAR(replace=True, x_ind=xi)._update_inputs(x, xi, y_hole, f, obs), expect([9, NAN, 11], [12, 13]), atol=1e-6)
    close(GPAR(impute=True, replace=True, x_ind=xi)._update_inputs(x, xi, y, f, obs), expect([9, 10, 11], [12, 13]), atol=1e-6)


def test_conditioning_interpolates_low_noise_data(engine, x, w):
    prior = Measure()
    f1, n1 = GP(EQ(), measure=prior), 1e-10
    f2, n2 = GP(EQ(), measure=prior), 2e-10
    gpar = GPAR().add_layer(lambda: (f1, n1)).add_layer(lambda: (f2, n2))
    y = torch.cat([f1(x, n1).sample(), f2(x, n2).sample()], dim=1)
    post = gpar | (x, y, w)
    p1, pn1 = post.layers[0]()
    p2, pn2 = post.layers[1]()
    assert pn1 == n1 and pn2 == n2
    close(p1.mean(x), y[:, 0:1], atol=1e-3)
    x2 = np.concatenate([x, to_np(y[:, 0:1])], axis=1)
    close(p2.mean(x2), y[:, 1:2], atol=1e-3)


def test_logpdf_is_sum_of_layer_logpdfs(engine, x, w):
    prior = Measure()
    f1, n1 = GP(EQ(), measure=prior), 2e-1
    f2, n2 = GP(Linear(), measure=prior), 1e-1
    gpar = GPAR().add_layer(lambda: (f1, n1)).add_layer(lambda: (f2, n2))
    y = gpar.sample(x, w, latent=True)
    x2 = np.concatenate([x, to_np(y[:, 0:1])], axis=1)
    l1 = f1(x, n1 / w[:, 0]).logpdf(y[:, 0])
    l2 = f2(x2, n2 / w[:, 1]).logpdf(y[:, 1])

    assert float(gpar.logpdf(x, y, w)) == float(l1 + l2)  # exact: same device computation, same order
    assert float(gpar.logpdf(x, y, w, only_last_layer=True)) == float(l2)

    # resume: inputs after layer 0, then only layer 1
    x_part, xi_part = gpar.logpdf(x, y, w, return_inputs=True, outputs=[0])
    assert float(gpar.logpdf(x_part, y, w, x_ind=xi_part, outputs=[1])) == float(l2)

    # sampling the missing value gives a stochastic estimate
    y = y.clone()
    y[1, 0] = NAN
    columns_all_different(gpar.logpdf(x, y, w, sample_missing=True), gpar.logpdf(x, y, w, sample_missing=True))


def test_samples_are_random_and_posterior_samples_hit_the_data(engine, x, w):
    prior = Measure()
    f1, f2 = GP(EQ(), measure=prior), GP(EQ(), measure=prior)
    gpar = GPAR().add_layer(lambda: (f1, 1e-1)).add_layer(lambda: (f2, 2e-1))
    columns_all_different(gpar.sample(x, w), gpar.sample(x, w))
    columns_all_different(gpar.sample(x, w, latent=True), gpar.sample(x, w, latent=True))

    gpar = GPAR().add_layer(lambda: (f1, 1e-10)).add_layer(lambda: (f2, 2e-10))
    y = gpar.sample(x, w, latent=True)
    post = gpar | (x, y, w)
    close(post.sample(x, w), y, atol=1e-3)
    close(post.sample(x, w, latent=True), y, atol=1e-3)


def test_missing_rows_with_imputation_chain(engine):
    """A data set that is not closed downwards: layer masks, imputation and the posterior chain agree with a
    by-hand computation on the kept rows (exercises GPAR.__or__ + _update_inputs with impute=True)."""
    rng = np.random.default_rng(5)
    n = 12
    x = rng.standard_normal((n, 1))
    y = rng.standard_normal((n, 2))
    y[[1, 4], 0] = NAN  # missing in output 0 but observed in output 1 -> must be imputed
    y[[2, 7], 1] = NAN
    w = np.ones((n, 2))
    prior = Measure()
    f1, f2 = GP(EQ(), measure=prior), GP(EQ().stretch(2.0), measure=prior)
    gpar = GPAR(impute=True).add_layer(lambda: (f1, 0.1)).add_layer(lambda: (f2, 0.2))
    total = gpar.logpdf(x, y, w)

    have0 = ~np.isnan(y[:, 0])
    l1 = f1(x[have0], 0.1).logpdf(y[have0, 0])
    post1 = f1 | (f1(x[have0], 0.1), y[have0, 0])
    col = y[:, 0:1].copy()
    col[~have0] = to_np(post1.mean(x[~have0]))
    x2 = np.concatenate([x, col], axis=1)
    have1 = ~np.isnan(y[:, 1])
    l2 = f2(x2[have1], 0.2).logpdf(y[have1, 1])
    close(total, l1 + l2, rtol=1e-10)


def test_cholesky_retry_factor_as_in_lab(engine):
    """lab's `B.cholesky_retry_factor` (default 1: a failed Cholesky raises; > 1: retried with the jitter x 10 while the
    factor stays below it).  A rank-deficient noise-free covariance with epsilon = 0 fails; with the retry ladder
    allowed to reach a workable jitter it goes through."""
    from gpar_amd.engine import NotPositiveDefiniteError
    from gpar_amd.gp import GP
    from gpar_amd.kernels import Linear

    f = GP(Linear().stretch(np.ones(1)))  # rank one: any three points give a singular 3 x 3 matrix
    x = np.array([[1.0], [2.0], [3.0]])
    y = np.array([1.0, 2.0, 3.1])
    previous = engine.epsilon, engine.cholesky_retry_factor
    try:
        engine.epsilon = 1e-30
        with pytest.raises(NotPositiveDefiniteError):
            float(f(x).logpdf(y))
        engine.cholesky_retry_factor = 1e25  # jitter may grow to 1e-5
        value = float(f(x).logpdf(y))
        assert np.isfinite(value)
        sample = f(x).sample()
        assert np.all(np.isfinite(sample.cpu().numpy()))
        engine.cholesky_retry_factor = 10.0  # one retry only (1e-29): still singular in fp64
        with pytest.raises(NotPositiveDefiniteError):
            float(f(x).logpdf(y))
    finally:
        engine.epsilon, engine.cholesky_retry_factor = previous
