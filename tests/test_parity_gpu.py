"""GPU parity proper: everything below runs through the C ABI (libgpar_hip.so) on cuda:0 and is compared with

  (a) the committed golden vectors (tests/golden/gpar_cases.json; closed-form CPU values),
  (b) the CPU oracle engine on the same seeded inputs at sizes the oracle finishes in seconds,
  (c) size-independent properties at the benchmark sizes (BASELINE.json configs): L L^T v = K v, the
      log-likelihood chain rule, VFE tightness at Z = X, Schur-complement identities.

Tolerances (fp64, stated per SURVEY.md §7(vi)): log marginal likelihood rtol 1e-10 against the oracle for
well-conditioned problems (noise >= 1e-2 of the signal); posterior moments rtol 1e-8 / atol 1e-10; samples with a
shared Philox stream atol 1e-8; trained hyper-parameters after identical L-BFGS-B iteration counts rtol 1e-5.
"""
import json

import numpy as np
import pytest
import torch

from .conftest import make_engine, to_np
from .test_oracle import GOLDEN, _kernel_from_spec, _nan_array, regressor_from_case

pytestmark = pytest.mark.gpu


@pytest.fixture
def hip():
    from gpar_amd.engine import set_engine

    eng = make_engine("hip")
    previous = set_engine(eng)
    yield eng
    set_engine(previous)


def _golden():
    with open(GOLDEN) as f:
        return json.load(f)


def _on(kind, fn, seed=77):
    """Run fn() with the given engine installed."""
    from gpar_amd.engine import set_engine

    eng = make_engine(kind, seed=seed)
    previous = set_engine(eng)
    try:
        return fn()
    finally:
        set_engine(previous)


@pytest.mark.parametrize("case", _golden()["gpar_logpdf"], ids=lambda c: c["name"])
def test_golden_gpar_logpdf(case, hip):
    x, y = np.array(case["x"]), _nan_array(case["y"])
    w = None if case["w"] is None else np.array(case["w"])
    hip.epsilon = case.get("epsilon", 1e-12)  # lab's B.epsilon; the air_temp workload sets 1e-6
    got = float(regressor_from_case(case).logpdf(x, y, w))
    # inducing-point chains feed posterior means through K_zz^-1 (jitter-conditioned): 1e-8, as the other VFE goldens
    tol = 1e-10 if case.get("x_ind") is None else 1e-8
    assert abs(got - case["logpdf"]) <= tol * abs(case["logpdf"]), (got, case["logpdf"])


@pytest.mark.parametrize("case", _golden()["single_gp"], ids=lambda c: c["name"])
def test_golden_posterior_moments(case, hip):
    from gpar_amd.gp import GP, Obs
    from oracle import gpar_ref

    spec, _ = gpar_ref.layer_spec(case["hypers"], 2, 0, case["config"])
    f = GP(_kernel_from_spec(spec))
    x, y, noise, xs = (np.array(case[k]) for k in ("x", "y", "noise", "xs"))
    assert abs(float(f(x, noise).logpdf(y)) - case["logpdf"]) <= 1e-10 * abs(case["logpdf"])
    post = f | Obs(f(x, noise), y)
    np.testing.assert_allclose(to_np(post.mean(xs))[:, 0], case["mean"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(to_np(post(xs).var()), case["cov"], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("case", _golden()["vfe"], ids=lambda c: c["name"])
def test_golden_inducing_points(case, hip):
    from gpar_amd.gp import GP, PseudoObs
    from oracle import gpar_ref

    spec, _ = gpar_ref.layer_spec(case["hypers"], 1, 0, case["config"])
    f = GP(_kernel_from_spec(spec))
    x, y, noise, z, xs = (np.array(case[k]) for k in ("x", "y", "noise", "z", "xs"))
    obs = PseudoObs(f(z), f(x, noise), y)
    assert abs(float(obs.logpdf()) - case["bound"]) <= 1e-8 * abs(case["bound"])
    post = f | obs
    np.testing.assert_allclose(to_np(post.mean(xs))[:, 0], case["mean"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(to_np(post(xs).var()), case["cov"], rtol=1e-6, atol=1e-8)


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
    y = (y - y.mean(0)) / y.std(0)
    if missing:
        y[rng.random(y.shape) < missing] = np.nan
    return x, y


CONFIGS = {
    "C1-paper-synthetic": (dict(scale=0.1, linear=True, linear_scale=10.0, nonlinear=True, nonlinear_scale=0.1, noise=0.1, impute=True, replace=False, normalise_y=False), 25, 1, 3, 0.0),
    "C2-shape": (dict(scale=0.5, linear=True, nonlinear=False, noise=0.1), 384, 2, 4, 0.0),
    "C3-shape-markov2": (dict(scale=0.5, linear=True, nonlinear=True, markov=2, noise=0.1), 300, 4, 8, 0.0),
    "C4-shape-inducing": (dict(scale=0.5, linear=True, nonlinear=True, noise=0.1, x_ind=np.random.default_rng(8).uniform(0, 1, (64, 8))), 500, 8, 4, 0.0),
    "C5-shape-per-rq": (dict(scale=0.5, per=True, rq=True, linear=True, nonlinear=True, noise=0.1), 200, 3, 5, 0.0),
    "missing-impute": (dict(scale=0.5, linear=True, nonlinear=True, noise=0.1, impute=True), 257, 2, 3, 0.2),
    "replace": (dict(scale=0.5, linear=True, nonlinear=True, noise=0.1, impute=True, replace=True), 190, 2, 3, 0.1),
}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_logpdf_condition_predict_match_oracle(name):
    from gpar_amd.regression import GPARRegressor

    kw, n, m, p, missing = CONFIGS[name]
    x, y = _problem(n, m, p, seed=len(name), missing=missing)
    xs = np.random.default_rng(1).uniform(0, 1, (40, m))

    def run():
        reg = GPARRegressor(**kw)
        prior = float(reg.logpdf(x, y))
        reg.condition(x, y)
        post = float(reg.logpdf(x, y, posterior=True)) if not missing else 0.0
        samples = reg.sample(xs, posterior=True, num_samples=3, latent=True)
        return prior, post, np.stack(samples)

    ref = _on("oracle", run)
    got = _on("hip", run)
    assert abs(got[0] - ref[0]) <= 1e-10 * abs(ref[0]), (got[0], ref[0])
    assert abs(got[1] - ref[1]) <= 1e-8 * max(1.0, abs(ref[1])), (got[1], ref[1])
    # Latent samples are mean + chol(cov + 1e-12 I) z.  Where the test inputs are dense relative to the length scale
    # (C1: 40 points, scale 0.1) the posterior covariance is numerically singular and the factor of the jittered
    # matrix is only determined to ~sqrt(eps_jitter) = 1e-6 per layer (amplified along the chain), on any hardware.
    atol = 5e-5 if name == "C1-paper-synthetic" else 1e-7
    np.testing.assert_allclose(got[2], ref[2], rtol=1e-6, atol=atol)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_logpdf_matches_the_independent_oracle_route(name, hip):
    """HIP against `oracle.gpar_ref.gpar_logpdf` directly: that restatement shares NOTHING with the product - its own layer
    kernels from the hyper-parameter dictionary, its own missing-data bookkeeping, slogdet + solve instead of a Cholesky of an
    augmented matrix - whereas the test above runs the product's host algebra on both engines.  (Before round 3 this route
    reached the HIP path only through the golden vectors, at n <= 30.)"""
    from gpar_amd.regression import GPARRegressor
    from oracle import gpar_ref

    kw, n, m, p, missing = CONFIGS[name]
    x, y = _problem(n, m, p, seed=len(name), missing=missing)
    reg = GPARRegressor(**dict(kw, normalise_y=False))
    got = float(reg.logpdf(x, y))
    want = gpar_ref.gpar_logpdf(x, y, None, reg.get_variables(), reg.model_config, impute=reg.impute, replace=reg.replace,
                                x_ind=kw.get("x_ind"))
    tol = 1e-10 if kw.get("x_ind") is None else 1e-8  # inducing-point chains go through K_zz^-1 (jitter-conditioned)
    assert abs(got - want) <= tol * abs(want), (got, want)


@pytest.mark.parametrize("kw", [
    dict(impute=True),
    dict(impute=False),
    dict(impute=True, replace=True),
    dict(impute=True, x_ind=np.random.default_rng(4).uniform(0, 1, (40, 2))),
], ids=["impute", "no-impute", "replace", "inducing"])
def test_sampled_imputations_match_the_oracle_on_a_shared_stream(kw):
    """`logpdf(sample_missing=True)` (reference gpar/model.py:229-237) with holes that are NOT closed downwards, n = 300, p = 4:
    the device draws its imputations from Philox stream (seed, call index), which the oracle restates bit for bit - same seed,
    same imputations, so the VALUES agree to rounding: against the product's host algebra on the numpy engine and against the
    independent restatement `oracle.gpar_ref.gpar_logpdf(sample_missing=True)`.  A different seed gives a different value."""
    from gpar_amd.regression import GPARRegressor
    from oracle import gpar_ref

    x, y = _problem(300, 2, 4, seed=31, missing=0.2)
    y[0] = 0.25  # one complete row
    assert np.isnan(y[:, 0]).any() and (np.isnan(y[:, 0]) & ~np.isnan(y[:, 2])).any()  # not closed downwards

    def run():
        reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1, normalise_y=False, **kw)
        return float(reg.logpdf(x, y, sample_missing=True)), reg.get_variables(), reg.model_config

    got, hypers, config = _on("hip", run, seed=41)
    ref, _, _ = _on("oracle", run, seed=41)
    other, _, _ = _on("hip", run, seed=42)
    independent = gpar_ref.gpar_logpdf(x, y, None, hypers, config, impute=kw.get("impute", True), replace=kw.get("replace", False),
                                       x_ind=kw.get("x_ind"), sample_missing=True, seed=41)
    tol = 1e-8 if kw.get("x_ind") is None else 1e-7
    assert abs(got - ref) <= tol * abs(ref), (got, ref)
    # (inducing points: the independent route solves against K_zz + 1e-12 I directly, the product through its Cholesky factor -
    # with 40 random inducing inputs at scale 0.5 that matrix is conditioned ~1e9, and the drawn imputations inherit the difference)
    assert abs(got - independent) <= (tol if kw.get("x_ind") is None else 1e-5) * abs(independent), (got, independent)
    if kw.get("replace"):
        # impute AND replace: `_update_inputs` overwrites the whole column by posterior means (gpar/model.py:305-306), the draws included
        assert other == got
    else:
        assert abs(other - got) > 1e-6 * abs(got)


def test_layer_pipeline_is_bit_identical_to_serial_evaluation(monkeypatch):
    """Independent layers run on alternating streams (HipEngine.pipeline); values and their summation order are
    those of the serial loop, so the result must not change by a single bit - at a size where the look-ahead path
    (second stream inside gpar_potrf) is active too, so three kinds of concurrency are in play."""
    from gpar_amd.model import GPAR
    from gpar_amd.parallel import sharded_logpdf
    from gpar_amd.regression import GPARRegressor, _construct_gpar

    x, y = _problem(7300, 3, 5, seed=21)

    def run():
        import torch

        from gpar_amd.engine import get_engine

        reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1)
        out = {}
        for depth in ["0", "2", "3"]:
            monkeypatch.setenv("GPAR_LAYER_PIPELINE", depth)
            out[depth] = float(reg.logpdf(x, y))
        eng = get_engine()
        gpar = _construct_gpar(reg, reg.vs, 3, 5)
        xd, yd = eng.tensor(x), eng.tensor(y)  # an unconditioned regressor applies identity transforms
        for depth in ["0", "2"]:
            monkeypatch.setenv("GPAR_LAYER_PIPELINE", depth)
            out["sharded" + depth] = float(sharded_logpdf(gpar, xd, yd, torch.ones_like(yd)))
        assert isinstance(gpar, GPAR)
        return out

    got = _on("hip", run)
    # (depth 0 factors the layers one at a time, depths 2 and 3 in lock-step at this size: the schedule of a factorisation - which
    # steps take the small-tile update kernel, which pairs of panels share a launch - is a function of the rows left AND of the
    # batch, so the two agree to rounding, and each is repeatable to the bit)
    assert got["2"] == got["3"] == got["sharded2"], got
    assert got["0"] == got["sharded0"], got
    assert abs(got["0"] - got["2"]) <= 1e-15 * abs(got["0"]), got


@pytest.mark.parametrize("n,p", [(2100, 4), (512, 3), (4096, 2), (5000, 3)])
def test_lockstep_layers_equal_layer_by_layer_evaluation(monkeypatch, n, p):
    """Small independent layers are factored together in lock-step (HipEngine.logpdf_dense_batch, up to GPAR_LAYER_BATCH_ROWS
    rows) instead of on separate streams: the same log marginal likelihood as the pipelined and the serial evaluation, and as
    the oracle where that is cheap."""
    from gpar_amd.regression import GPARRegressor

    x, y = _problem(n, 2, p, seed=n)

    def run():
        reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1)
        out = {}
        out["lockstep"] = float(reg.logpdf(x, y))
        monkeypatch.setenv("GPAR_LAYER_BATCH_BYTES", str(8 * (n + 1) * (n + 17) * 2))   # two layers per batch (a last one alone)
        out["lockstep_chunks"] = float(reg.logpdf(x, y))
        monkeypatch.delenv("GPAR_LAYER_BATCH_BYTES")
        monkeypatch.setenv("GPAR_LAYER_BATCH_ROWS", "0")
        out["streams"] = float(reg.logpdf(x, y))
        monkeypatch.setenv("GPAR_LAYER_PIPELINE", "0")
        out["serial"] = float(reg.logpdf(x, y))
        monkeypatch.delenv("GPAR_LAYER_PIPELINE")
        monkeypatch.delenv("GPAR_LAYER_BATCH_ROWS")
        return out

    got = _on("hip", run)
    assert got["streams"] == got["serial"], got
    assert abs(got["lockstep"] - got["serial"]) <= 1e-11 * abs(got["serial"]), got
    assert abs(got["lockstep_chunks"] - got["serial"]) <= 1e-11 * abs(got["serial"]), got
    if n <= 600:
        ref = _on("oracle", lambda: float(GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1).logpdf(x, y)))
        assert abs(got["lockstep"] - ref) <= 1e-9 * abs(ref), (got, ref)


@pytest.mark.parametrize("kw,latent", [(dict(), True), (dict(), False), (dict(markov=1), True), (dict(input_linear=True, rq=True), True),
                                       (dict(linear_scale=3.0, scale=0.3), False), (dict(nonlinear=True), True)])
def test_linear_output_dependence_samples_with_one_shared_solve(monkeypatch, kw, latent):
    """The reference's DEFAULT output dependence (linear=True, nonlinear=False: gpar/regression.py:141-146, 276-278) makes the
    sample-dependent part of K(X, x*_s) a rank-q matrix; gp.Obs._sample_batch_linear_tail then draws all samples of a layer from
    ONE triangular solve and ONE rank-n downdate plus a rank-2q update per sample.  Same random numbers in, the same samples out
    as the general routine (GPAR_LINEAR_TAIL=0) up to rounding; a kernel with a nonlinear output part does not qualify (same bits)."""
    from gpar_amd.engine import get_engine
    from gpar_amd.regression import GPARRegressor

    n, m, p, S, ns = 900, 2, 4, 7, 300
    x, y = _problem(n, m, p, seed=5)
    xs = np.random.default_rng(6).uniform(0, 1, (ns, m))
    w = np.random.default_rng(7).uniform(0.5, 2.0, (ns, p))

    def run():
        out = {}
        for mode in ["1", "0"]:
            monkeypatch.setenv("GPAR_LINEAR_TAIL", mode)
            reg = GPARRegressor(**dict(dict(scale=0.5, linear=True, nonlinear=False, noise=0.1), **kw))
            reg.condition(x, y)
            get_engine().seed(33)
            out[mode] = np.stack(reg.sample(xs, w=w, posterior=True, num_samples=S, latent=latent))
        monkeypatch.delenv("GPAR_LINEAR_TAIL")
        return out

    got = _on("hip", run)
    assert got["1"].shape == (S, ns, p) and np.all(np.isfinite(got["1"]))
    if kw.get("nonlinear"):
        assert np.array_equal(got["1"], got["0"])
    else:
        assert not np.array_equal(got["1"], got["0"])   # (the other routine did run)
        # With observation noise in the covariance the two routines agree to rounding.  A LATENT draw factors K** - V^T V + 1e-12 I,
        # which is numerically singular at 300 points: two algebraically equal ways of forming it differ by ~1e-11, and a draw
        # moves by that over sqrt(jitter) in the directions the data pin down - the same for any reordering of the general
        # routine's sums.  The law is the same; the tolerance says what "the same samples" can mean there.
        err = np.max(np.abs(got["1"] - got["0"]))
        assert err < (1e-4 if latent else 1e-10), err


@pytest.mark.parametrize("n,p,weights", [(700, 4, False), (512, 3, True), (1300, 5, True), (40, 70, True), (2100, 2, False)])
def test_one_call_lockstep_evaluation_returns_the_same_bits(monkeypatch, n, p, weights):
    """gpar_logpdf_lockstep (ABI v5): the whole lock-step evaluation in one library call - no design matrix, noise tensor or
    observation object per layer - returns bit for bit what the per-layer build calls return (GPAR_ONE_CALL=0), with weights
    (noise / w divided on the device), with more layers than one launch of the preparing kernel takes (70 > 64), in chunks,
    and for `only_last_layer`."""
    import torch

    from gpar_amd.engine import get_engine
    from gpar_amd.regression import GPARRegressor, _construct_gpar

    x, y = _problem(n, 2, min(p, 6), seed=n + p)
    if p > 6:
        y = np.concatenate([y] * (p // y.shape[1] + 1), axis=1)[:, :p] + 0.01 * np.random.default_rng(0).standard_normal((n, p))
    w = np.random.default_rng(1).uniform(0.5, 2.0, y.shape) if weights else None

    def run():
        eng = get_engine()
        kw = dict(scale=0.5, linear=True, nonlinear=True, noise=0.1, markov=3 if p > 6 else None, normalise_y=False)
        out = {}
        for mode in ["1", "0"]:
            monkeypatch.setenv("GPAR_ONE_CALL", mode)
            reg = GPARRegressor(**kw)
            out[mode] = float(reg.logpdf(x, y, w))
            gpar = _construct_gpar(reg, reg.vs, 2, p)
            wd = eng.tensor(np.ones_like(y) if w is None else w)
            out[mode + "last"] = float(gpar.logpdf(eng.tensor(x), eng.tensor(y), wd, only_last_layer=True))
            monkeypatch.setenv("GPAR_LAYER_BATCH_BYTES", str(8 * (n + 1) * (n + 17) * 3))   # three layers per batch
            out[mode + "chunks"] = float(reg.logpdf(x, y, w))
            monkeypatch.delenv("GPAR_LAYER_BATCH_BYTES")
        monkeypatch.delenv("GPAR_ONE_CALL")
        return out

    got = _on("hip", run)
    assert got["1"] == got["0"] and got["1last"] == got["0last"] and got["1chunks"] == got["0chunks"], got
    if n <= 600:
        ref = _on("oracle", lambda: float(GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1, markov=3 if p > 6 else None,
                                                        normalise_y=False).logpdf(x, y, w)))
        assert abs(got["1"] - ref) <= 1e-9 * abs(ref), (got, ref)


@pytest.mark.parametrize("n,p", [(1500, 4), (300, 3)])
def test_lockstep_conditioning_equals_layer_by_layer(monkeypatch, n, p):
    """Conditioning on complete data factors the (independent) layers in lock-step (HipEngine.factor_dense_batch): the posterior
    it yields - read through the posterior log-density of held-out data and a posterior sample with a fixed seed - is the one
    the layer-by-layer factorisations yield."""
    from gpar_amd.engine import get_engine
    from gpar_amd.regression import GPARRegressor

    x, y = _problem(n + 200, 2, p, seed=n)
    xt, yt, xh, yh = x[:n], y[:n], x[n:], y[n:]

    def run():
        out = {}
        for mode in ["lockstep", "streams", "serial"]:
            if mode == "streams":
                monkeypatch.setenv("GPAR_LAYER_BATCH_ROWS", "0")
            if mode == "serial":
                monkeypatch.setenv("GPAR_LAYER_PIPELINE", "0")
            reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1)
            reg.condition(xt, yt)
            out[mode] = float(reg.logpdf(xh, yh, posterior=True))
            get_engine().seed(5)
            out[mode + "_sample"] = np.asarray(reg.sample(xh[:50], posterior=True, latent=True))
            get_engine().seed(6)   # several samples: the per-sample blocks of every layer go through the batched kernels
            out[mode + "_samples"] = np.stack([np.asarray(v) for v in reg.sample(xh[:130], posterior=True, num_samples=5)])
        monkeypatch.delenv("GPAR_LAYER_PIPELINE")
        monkeypatch.delenv("GPAR_LAYER_BATCH_ROWS")
        return out

    got = _on("hip", run)
    assert got["streams"] == got["serial"], got
    assert abs(got["lockstep"] - got["serial"]) <= 1e-10 * abs(got["serial"]), got
    np.testing.assert_allclose(got["lockstep_sample"], got["serial_sample"], rtol=1e-7, atol=1e-8)
    assert got["lockstep_samples"].shape == (5, 130, p)
    np.testing.assert_allclose(got["lockstep_samples"], got["serial_samples"], rtol=1e-6, atol=1e-7)


def test_concurrent_layer_training_equals_serial_training(monkeypatch):
    """fit(fix=True) on observed data trains independent layers from two host threads on two streams; every objective
    evaluation is the same deterministic device computation, so the trained hyper-parameters are those of the serial
    loop, bit for bit.  With a shared hyper-parameter (scale_tie) the layers are NOT independent and fit stays serial."""
    from gpar_amd.parallel import layers_train_independently
    from gpar_amd.regression import GPARRegressor

    x, y = _problem(900, 2, 4, seed=12)

    def run():
        out = {}
        for threads in ["1", "2"]:
            monkeypatch.setenv("GPAR_FIT_THREADS", threads)
            reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1)
            reg.fit(x, y, iters=4)
            out[threads] = {k: np.array(v) for k, v in reg.get_variables().items()}
        tied = GPARRegressor(scale=0.5, scale_tie=True, linear=True, nonlinear=True, noise=0.1)
        tied.condition(x, y)
        import torch

        out["tied_independent"] = layers_train_independently(tied, torch.as_tensor(y))
        return out

    got = _on("hip", run)
    assert got["1"].keys() == got["2"].keys() and len(got["1"]) > 8
    for name in got["1"]:
        assert np.array_equal(got["1"][name], got["2"][name]), name
    assert got["tied_independent"] is False


def test_default_thread_count_of_fit_follows_the_problem_size(monkeypatch):
    """HipEngine.worker_streams: four training threads from 2048 to 5120 rows, two otherwise (profiles/r04_fit_threads.txt);
    at a size where the default is four, the trained hyper-parameters are those of the serial loop, bit for bit."""
    from gpar_amd.regression import GPARRegressor

    x, y = _problem(2100, 2, 4, seed=13)

    def run():
        from gpar_amd.engine import get_engine

        eng = get_engine()
        monkeypatch.delenv("GPAR_FIT_THREADS", raising=False)
        counts = {rows: len(eng.worker_streams(rows=rows)) for rows in (400, 2047, 2048, 5120, 5121, 16384)}
        out = {"counts": counts}
        for threads in [None, "1"]:
            if threads is not None:
                monkeypatch.setenv("GPAR_FIT_THREADS", threads)
            reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1)
            reg.fit(x, y, iters=3)
            out[threads] = {k: np.array(v) for k, v in reg.get_variables().items()}
        return out

    got = _on("hip", run)
    assert got["counts"] == {400: 2, 2047: 2, 2048: 4, 5120: 4, 5121: 2, 16384: 2}
    for name in got["1"]:
        assert np.array_equal(got[None][name], got["1"][name]), name


def test_predict_reduction_on_device_matches_numpy_reduction():
    """predict = device-side mean / percentiles of the same samples `sample` returns (row a10 of SURVEY section 8):
    same seed -> the HIP reduction equals numpy's on the HIP samples bit for bit, and the oracle's to sample parity."""
    from gpar_amd.engine import get_engine
    from gpar_amd.regression import GPARRegressor

    x, y = _problem(150, 2, 3, seed=9)
    xs = np.random.default_rng(4).uniform(0, 1, (33, 2))

    def run():
        reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1)
        reg.condition(x, y)
        get_engine().seed(123)
        samples = np.stack(reg.sample(xs, posterior=True, num_samples=40))
        get_engine().seed(123)
        mean, lo, hi = reg.predict(xs, num_samples=40, credible_bounds=True)
        get_engine().seed(123)
        only_mean = reg.predict(xs, num_samples=40)
        return samples, mean, lo, hi, only_mean

    got = _on("hip", run)
    assert np.array_equal(got[1], np.mean(got[0], axis=0))
    assert np.array_equal(got[2], np.percentile(got[0], 2.5, axis=0))
    assert np.array_equal(got[3], np.percentile(got[0], 97.5, axis=0))
    assert np.array_equal(got[4], got[1])
    ref = _on("oracle", run)
    for g, r in zip(got[1:], ref[1:]):
        np.testing.assert_allclose(g, r, rtol=1e-6, atol=1e-7)


def test_sparse_path_matches_oracle():
    from gpar_amd.regression import GPARRegressor

    x, y = _problem(600, 2, 3, seed=4)
    z = np.random.default_rng(2).uniform(0, 1, (48, 2))
    xs = np.random.default_rng(3).uniform(0, 1, (30, 2))

    def run():
        reg = GPARRegressor(x_ind=z, scale=0.5, linear=True, nonlinear=True, noise=0.1)
        bound = float(reg.logpdf(x, y))
        reg.condition(x, y)
        return bound, np.stack(reg.sample(xs, posterior=True, num_samples=2, latent=True))

    ref, got = _on("oracle", run), _on("hip", run)
    assert abs(got[0] - ref[0]) <= 1e-9 * abs(ref[0])
    np.testing.assert_allclose(got[1], ref[1], rtol=1e-5, atol=1e-6)


def test_fit_matches_oracle_training():
    from gpar_amd.regression import GPARRegressor

    x, y = _problem(120, 1, 3, seed=12)

    def run():
        reg = GPARRegressor(scale=0.3, linear=True, nonlinear=True, nonlinear_scale=0.5, noise=0.3)
        reg.fit(x, y, iters=12)
        return reg.get_variables(), float(reg.logpdf(x, y))

    (vr, lr), (vg, lg) = _on("oracle", run), _on("hip", run)
    assert sorted(vr) == sorted(vg)
    for k in vr:
        np.testing.assert_allclose(vg[k], vr[k], rtol=1e-5, atol=1e-8, err_msg=k)
    assert abs(lg - lr) <= 1e-7 * abs(lr)


def test_gradient_matches_oracle(hip):
    from gpar_amd.regression import GPARRegressor

    x, y = _problem(150, 2, 3, seed=21)

    def grads():
        reg = GPARRegressor(scale=0.5, per=True, rq=True, input_linear=True, linear=True, nonlinear=True, noise=0.1, normalise_y=False, impute=False)
        with torch.no_grad():
            reg.logpdf(x, y)
        reg.vs.requires_grad(True)
        reg.logpdf(torch.tensor(x), torch.tensor(y)).backward()
        return np.concatenate([v.grad.numpy().reshape(-1) for v in reg.vs.get_vars()])

    ref, got = _on("oracle", grads), _on("hip", grads)
    np.testing.assert_allclose(got, ref, rtol=1e-8, atol=1e-9 * np.max(np.abs(ref)))


@pytest.mark.parametrize("kw", [
    dict(impute=True, replace=True, x_ind=np.linspace(0, 1, 24)),
    dict(impute=True, replace=True),
    dict(impute=True, replace=True, x_ind=np.linspace(0, 1, 24), sparse_method="fitc"),
], ids=["impute+replace+inducing", "impute+replace", "impute+replace+fitc"])
def test_joint_gradient_matches_oracle(hip, kw):
    """The JOINT objective of fit(fix=False) (reference gpar/regression.py:447-456) in regimes where posterior means are fed
    forward - imputed AND replaced columns, inducing inputs extended layer by layer: d / d(every hyper-parameter) through
    `_PosteriorMean` and the input gradients of `_LogMarginal`, HIP kernels against the numpy engine."""
    from gpar_amd.regression import GPARRegressor

    rng = np.random.default_rng(19)
    n, p = 180, 3
    x = np.sort(rng.uniform(0, 1, n))
    y = np.stack([np.sin(5 * x), np.cos(4 * x) + 0.4 * np.sin(5 * x) ** 2, x * np.sin(5 * x)], axis=1) + 0.05 * rng.standard_normal((n, p))
    y[rng.random((n, p)) < 0.2] = np.nan
    y[0] = [0.1, 0.9, 0.0]

    def grads():
        reg = GPARRegressor(scale=0.3, linear=True, linear_scale=3.0, nonlinear=True, rq=True, noise=0.05, normalise_y=False, **kw)
        with torch.no_grad():
            reg.logpdf(x, y)
        reg.vs.requires_grad(True)
        reg.logpdf(torch.tensor(x), torch.tensor(y)).backward()
        return np.concatenate([(v.grad if v.grad is not None else torch.zeros_like(v)).numpy().reshape(-1) for v in reg.vs.get_vars()])

    ref, got = _on("oracle", grads), _on("hip", grads)
    assert np.max(np.abs(ref)) > 1e-2
    np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-7 * np.max(np.abs(ref)))


def test_gradient_matches_oracle_at_a_size_that_takes_the_recursive_inverse(hip):
    """n = 1024 (a multiple of 512): K^-1 behind the analytic gradient comes from the recursive blocked inversion
    (batched diagonal blocks + one level of triangular-aware products), and every factorisation is two fused panels."""
    from gpar_amd.regression import GPARRegressor

    x, y = _problem(1024, 2, 2, seed=33)

    def grads():
        reg = GPARRegressor(scale=0.5, rq=True, linear=True, nonlinear=True, noise=0.1, normalise_y=False, impute=False)
        with torch.no_grad():
            reg.logpdf(x, y)
        reg.vs.requires_grad(True)
        reg.logpdf(torch.tensor(x), torch.tensor(y)).backward()
        return np.concatenate([v.grad.numpy().reshape(-1) for v in reg.vs.get_vars()])

    ref, got = _on("oracle", grads), _on("hip", grads)
    np.testing.assert_allclose(got, ref, rtol=1e-8, atol=1e-9 * np.max(np.abs(ref)))


def test_sparse_gradient_and_training_match_oracle(hip):
    """Inducing-point (VFE) path under training: gradient of the bound with respect to every hyper-parameter, and the
    hyper-parameters after a short L-BFGS-B run, HIP vs oracle (all kernel families; 257 points so the cross-gradient pass
    has ragged tiles; non-trivial weights through missing-free data and `w`)."""
    from gpar_amd.regression import GPARRegressor

    x, y = _problem(257, 2, 3, seed=23)
    z = np.random.default_rng(5).uniform(0, 1, (37, 2))
    w = np.random.default_rng(6).uniform(0.5, 1.5, y.shape)
    kw = dict(x_ind=z, scale=0.5, per=True, rq=True, input_linear=True, linear=True, nonlinear=True, noise=0.1, normalise_y=False)

    def grads():
        reg = GPARRegressor(**kw)
        with torch.no_grad():
            reg.logpdf(x, y, w)
        reg.vs.requires_grad(True)
        reg.logpdf(torch.tensor(x), torch.tensor(y), torch.tensor(w)).backward()
        return np.concatenate([(v.grad if v.grad is not None else torch.zeros_like(v)).numpy().reshape(-1) for v in reg.vs.get_vars()])

    def train():
        reg = GPARRegressor(**dict(kw, per=False, rq=False))
        reg.fit(x, y, w, iters=6)
        return reg.get_variables(), float(reg.logpdf(x, y, w))

    ref, got = _on("oracle", grads), _on("hip", grads)
    assert np.max(np.abs(ref)) > 1e-3
    np.testing.assert_allclose(got, ref, rtol=1e-7, atol=1e-8 * np.max(np.abs(ref)))
    (vr, lr), (vg, lg) = _on("oracle", train), _on("hip", train)
    assert sorted(vr) == sorted(vg)
    for k in vr:
        np.testing.assert_allclose(vg[k], vr[k], rtol=1e-4, atol=1e-7, err_msg=k)
    assert abs(lg - lr) <= 1e-6 * abs(lr)


# ---- properties at benchmark sizes (no oracle: too large for the CPU) -----------------------------------------

@pytest.mark.parametrize("n,m,p_cols", [(4096, 2, [2, 3]), (16384, 4, [9, 10]), (20011, 3, [4])])
def test_full_size_cholesky_properties(hip, n, m, p_cols):
    """K v = L (L^T v) and the augmented row equals L^-1 y, on the layer kernel at BASELINE sizes - and beyond them at
    a ragged n with more 64-row blocks than CUs (panel workgroups then own several row blocks each)."""
    from gpar_amd import hip as H
    from gpar_amd.kernels import EQ, Linear, compile_kernel

    dev = hip.device
    width = max(p_cols) + 1
    g = torch.Generator().manual_seed(n)
    x = torch.rand(n, width, generator=g, dtype=torch.float64).to(dev)
    k = (1.0 * EQ().stretch(np.full(m, 0.5))).select(list(range(m))) + (
        Linear().stretch(np.full(len(p_cols), 100.0)) + 1.0 * EQ().stretch(np.ones(len(p_cols)))
    ).select(p_cols)
    ck = compile_kernel(k, width)
    z = H.featurize(ck, x)
    A = H.alloc_matrix(n + 1, n + 1, dev)
    H.gram(ck, z, None, out=A[:n, :n], lower=True, diag_const=0.1 + 1e-12)
    yv = torch.randn(n, generator=g, dtype=torch.float64).to(dev)
    A[n, :n] = yv
    A[n, n] = 0.0
    v = torch.randn(n, 3, generator=g, dtype=torch.float64).to(dev)
    Kl = torch.tril(A[:n, :n])
    Kv = Kl @ v + torch.tril(Kl, -1).T @ v
    del Kl
    logdet, info = H.potrf_(A, nf=n)
    assert int(info.item()) == 0
    L = torch.tril(A[:n, :n])
    LLv = L @ (L.T @ v)
    rel = (LLv - Kv).norm() / Kv.norm()
    assert rel < 1e-12, rel
    # augmented row: z = L^-1 y  <=>  L z = y ; corner = -|z|^2
    zrow = A[n, :n]
    assert ((L @ zrow) - yv).norm() / yv.norm() < 1e-10
    assert abs(float(-A[n, n]) - float(zrow @ zrow)) <= 1e-10 * float(zrow @ zrow)
    assert abs(float(logdet) - 2 * float(torch.log(torch.diagonal(L)).sum())) <= 1e-10 * abs(float(logdet))


@pytest.mark.parametrize("n,bad", [(10000, 7777), (16384, 100), (16384, 16000)])
def test_non_positive_pivot_is_reported_at_full_size(hip, n, bad):
    """LAPACK-style info through every fast path at once (fused panels, paired panels, look-ahead on the side stream):
    the first non-positive pivot is reported 1-based, whatever garbage the later panels then compute."""
    from gpar_amd import hip as H

    dev = hip.device
    g = torch.Generator().manual_seed(1)
    x = torch.rand(n, 3, generator=g, dtype=torch.float64).to(dev)
    A = H.alloc_matrix(n, n, dev)
    A.copy_(torch.exp(-0.5 * torch.cdist(x, x) ** 2 / 0.25))
    A.diagonal().add_(0.1)
    A[bad, bad] = -5.0
    _, info = H.potrf_(A)
    assert int(info.item()) == bad + 1


def test_c2_logpdf_chain_rule_and_vfe_tightness(hip):
    """n = 4096, m = 2, p = 4 (BASELINE C2): the joint log-likelihood equals the sum of the per-layer conditionals
    computed one at a time, and with inducing points at a subset == all inputs of a small problem the bound is tight."""
    from gpar_amd.gp import GP, PseudoObs
    from gpar_amd.kernels import EQ
    from gpar_amd.regression import GPARRegressor, _construct_gpar

    x, y = _problem(4096, 2, 4, seed=2)
    reg = GPARRegressor(scale=0.5, linear=True, nonlinear=False, noise=0.1, normalise_y=False)
    total = float(reg.logpdf(x, y))
    gpar = _construct_gpar(reg, reg.vs, 2, 4)
    parts = 0.0
    for i in range(4):
        f, noise = gpar.layers[i]()
        xi = np.concatenate([x, y[:, :i]], axis=1)
        parts += float(f(xi, float(noise)).logpdf(y[:, i]))
    assert abs(total - parts) <= 1e-12 * abs(total)
    xs, ys = x[:700], y[:700, 0]
    f = GP(1.0 * EQ().stretch(np.full(2, 0.5)))
    exact = float(f(xs, 0.1).logpdf(ys))
    tight = float(PseudoObs(f(xs), f(xs, 0.1), ys).logpdf())
    assert abs(exact - tight) <= 1e-6 * abs(exact)


def _logpdf_longdouble(K, y):
    """log N(y; 0, K) by a Cholesky factorisation in 80-bit extended precision (numpy longdouble: 64-bit mantissa): the yardstick
    against which the fp64 implementations' rounding is measured."""
    A = np.array(K, dtype=np.longdouble)
    n = A.shape[0]
    L = np.zeros_like(A)
    for j in range(n):
        d = A[j, j] - np.dot(L[j, :j], L[j, :j])
        L[j, j] = np.sqrt(d)
        L[j + 1:, j] = (A[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    z = np.array(y, dtype=np.longdouble)
    for j in range(n):
        z[j] = (z[j] - np.dot(L[j, :j], z[:j])) / L[j, j]
    return -(np.sum(np.log(np.diag(L))) + np.longdouble(0.5) * n * np.log(2 * np.longdouble(np.pi)) + np.longdouble(0.5) * np.dot(z, z))


@pytest.mark.parametrize("n,noise", [(300, 0.1), (640, 0.01), (1100, 1e-3)])
def test_logpdf_rounding_against_extended_precision(hip, n, noise):
    """north_star asks for "ULP-level on logpdf".  One layer (EQ + linear kernel on two inputs), the SAME fp64 kernel matrix handed
    to (i) an 80-bit Cholesky, (ii) LAPACK in fp64, (iii) the HIP factorisation: the HIP value's distance from the extended-precision
    one is within four times LAPACK's plus 64 ulp of the value (measured: 46 / 175 / 175 ulp against LAPACK's 14 / 525 / 1119 at
    n = 300 / 640 / 1100 - sums of n logarithms and n squares in fp64 either way) - fused panels, look-ahead and all; and the whole product path
    (its own Gram build, with its own exponential) stays within 1e-13 relative."""
    from gpar_amd import hip as H
    from gpar_amd.gp import GP
    from gpar_amd.kernels import EQ, Linear

    rng = np.random.default_rng(n)
    x = rng.uniform(0, 1, (n, 2))
    y = np.sin(5 * x[:, 0]) + x[:, 1] + np.sqrt(noise) * rng.standard_normal(n)
    kernel = 1.3 * EQ().stretch(np.array([0.3, 0.6])) + 0.5 * Linear().stretch(np.array([2.0, 1.5]))
    from oracle import kernels as ok

    K = ok.gram(ok.spec_to_dict(kernel.resolve(2)), x, None, noise_diag=np.full(n, noise), jitter=1e-12)
    exact = _logpdf_longdouble(K, y)
    # (ii) LAPACK
    import scipy.linalg as sl

    Lr = sl.cholesky(K, lower=True)
    zr = sl.solve_triangular(Lr, y, lower=True)
    lapack = -(np.sum(np.log(np.diag(Lr))) + 0.5 * n * np.log(2 * np.pi) + 0.5 * zr @ zr)
    # (iii) the HIP factorisation of the same matrix: augmented [[K, .], [y^T, 0]] as the product does it
    dev = hip.device
    A = H.alloc_matrix(n + 1, n + 1, dev, zero=True)
    A[:n, :n] = torch.tensor(K, device=dev)
    A[n, :n] = torch.tensor(y, device=dev)
    logdet, info = H.potrf_(A, nf=n)
    assert int(info.item()) == 0
    quad = float((A[n, :n] ** 2).sum())
    mine = -(0.5 * float(logdet) + 0.5 * n * np.log(2 * np.pi) + 0.5 * quad)
    err_lapack, err_mine = abs(float(lapack - exact)), abs(float(mine - exact))
    ulp = np.spacing(abs(float(exact)))
    print(f"n={n} noise={noise}: |HIP - exact| = {err_mine / ulp:.1f} ulp, |LAPACK - exact| = {err_lapack / ulp:.1f} ulp")
    assert err_mine <= 4 * err_lapack + 64 * ulp, (err_mine / ulp, err_lapack / ulp)
    # the product path end to end
    f = GP(kernel)
    got = float(f(x, np.full(n, noise)).logpdf(y))
    print(f"   product path: {abs(got - float(exact)) / ulp:.1f} ulp")
    assert abs(got - float(exact)) <= 1e-13 * abs(float(exact)) + 4 * err_lapack, (got, float(exact))


def test_nan_pattern_of_device_outputs_is_remembered_until_they_change(hip):
    """GPAR._prep fetches the NaN pattern of device-resident outputs once per tensor VERSION (model._device_nan_pattern): the same
    tensor evaluated twice costs one fetch, an in-place write (here: an entry becomes missing) is seen at the next call."""
    import torch

    from gpar_amd import model
    from gpar_amd.regression import GPARRegressor

    x, y = _problem(300, 1, 3, seed=9)
    xd, yd = hip.tensor(x), hip.tensor(y)
    reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1, normalise_y=False)
    a = float(reg.logpdf(xd, yd))
    cached = yd._gpar_nan[1]   # (kept on the tensor object itself: nothing global holds the tensor or its pattern)
    assert float(reg.logpdf(xd, yd)) == a and yd._gpar_nan[1] is cached
    yd[5, 1] = float("nan")
    b = float(reg.logpdf(xd, yd))
    assert yd._gpar_nan[1] is not cached and np.isfinite(b) and b != a
    assert not hasattr(model, "_LAST_PATTERN")
    y2 = y.copy()
    y2[5, 1] = np.nan
    assert abs(b - float(reg.logpdf(x, y2))) <= 1e-12 * abs(b)


@pytest.mark.parametrize("kw,n", [(dict(linear=True, nonlinear=True), 500), (dict(per=True, rq=True, linear=True, nonlinear=True), 300),
                                  (dict(linear=True, nonlinear=False, input_linear=True), 1300)])
def test_one_call_objective_and_gradient_return_the_same_bits(monkeypatch, kw, n):
    """gpar_logpdf_dense_grad (ABI v5): a layer's training objective and the ingredients of its gradient in one library call -
    the launches of the separate entry points in their order, so value and gradient are bit for bit those of the two-step route
    (GPAR_ONE_CALL_GRAD_ROWS=0), with weights, periodic features (frequency derivatives) and a whole fit."""
    import torch

    from gpar_amd.regression import GPARRegressor

    x, y = _problem(n, 2, 3, seed=n)
    w = np.random.default_rng(3).uniform(0.5, 2.0, y.shape)

    def run():
        out = {}
        for mode in ["4096", "0"]:
            monkeypatch.setenv("GPAR_ONE_CALL_GRAD_ROWS", mode)
            reg = GPARRegressor(scale=0.5, noise=0.1, normalise_y=False, **kw)
            reg.logpdf(x, y, w)
            reg.vs.requires_grad(True)
            value = reg.logpdf(torch.tensor(x), torch.tensor(y), torch.tensor(w))
            value.backward()
            out[mode] = (float(value), {k: v.grad.clone().numpy() for k, v in zip(reg.vs.names, reg.vs.get_vars())})
            reg2 = GPARRegressor(scale=0.5, noise=0.1, **kw)
            reg2.fast_fit = False   # (the general route on both sides: this test compares the one-call and the two-step DEVICE routes;
            #                          the prepared objective against the general route is tests/test_fastfit.py)
            reg2.fit(x, y, w, iters=3)
            out[mode + "fit"] = reg2.get_variables()
        monkeypatch.delenv("GPAR_ONE_CALL_GRAD_ROWS")
        return out

    got = _on("hip", run)
    assert got["4096"][0] == got["0"][0]
    for name, g in got["4096"][1].items():
        assert np.array_equal(g, got["0"][1][name]), name
    for name, v in got["4096fit"].items():
        assert np.array_equal(v, got["0fit"][name]), name


def test_greedy_order_is_the_same_on_both_engines():
    """`GPARRegressor.greedy_order` (the reference's unimplemented greedy search, gpar/regression.py:400, 409-410) on the HIP engine
    - candidates of a position trained concurrently on the worker streams - and on the numpy engine: the same order and, after the
    same number of L-BFGS-B iterations from the same initial values, the same trained layer likelihoods (rtol 1e-6; the iterates
    agree to rounding, `test_fit_matches_oracle_training`)."""
    from gpar_amd.regression import GPARRegressor

    x = np.linspace(0, 1, 200)
    f1 = -np.sin(10 * np.pi * (x + 1)) / (2 * x + 1) - x**4
    f2 = np.cos(f1) ** 2 + np.sin(3 * x)
    f3 = f2 * f1**2 + 3 * x
    y = np.stack([f3, f1, f2], axis=1) + 0.1 * np.random.default_rng(1).standard_normal((200, 3))
    x, y = x[::2], y[::2]

    def run():
        reg = GPARRegressor(scale=0.1, linear=True, linear_scale=10.0, nonlinear=True, nonlinear_scale=0.1, noise=0.1, normalise_y=False)
        return reg.greedy_order(x, y, iters=15)

    ref, got = _on("oracle", run), _on("hip", run)
    assert got[0] == ref[0] == [1, 2, 0]
    np.testing.assert_allclose(got[1], ref[1], rtol=1e-6)
