"""The prepared training objective of a dense layer (gpar_amd/fastfit.py; reference gpar/regression.py:418-459, the objective
`minimise_l_bfgs_b` is handed per layer with fix=True).

CPU part: the HOST logic - the kernel expression traced over value holders, the bounded transforms, the mapping of the device
pass's per-position gradients to the store's variables, the chain rule in numpy - against the general route (torch autograd
through `GPAR.logpdf`) on the numpy engine, with the device side of the objective replaced by a dense numpy computation.
GPU part (`-m gpu`): the real thing through libgpar_hip.so - value to the bit, gradient to rounding, trained hyper-parameters
of `fit` equal on both routes."""
import numpy as np
import pytest
import torch

from gpar_amd import fastfit
from gpar_amd.engine import set_engine
from gpar_amd.model import per_output
from gpar_amd.optimise import objective_and_gradient
from gpar_amd.regression import GPARRegressor, _construct_gpar

from .conftest import make_engine

_LOG_2PI = float(np.log(2.0 * np.pi))


class NumpyObjective(fastfit.DenseLayerObjective):
    """DenseLayerObjective with its device side restated densely in numpy (test infrastructure: oracle kernels)."""

    def _allocate(self, ck):
        pass

    def _device_eval(self, ck, noise):
        from oracle import kernels as ok

        spec = ok.spec_to_dict(self.kernel.resolve(self.width))
        X, y, w = self.X.numpy(), self.y.numpy(), self.w.numpy()
        n = X.shape[0]
        K = ok.gram(spec, X, None, noise_diag=noise / w, jitter=self.eng.epsilon)
        try:
            L = np.linalg.cholesky(K)
        except np.linalg.LinAlgError:
            return None
        alpha = np.linalg.solve(K, y)
        value = -0.5 * (2.0 * np.sum(np.log(np.diag(L))) + n * _LOG_2PI + float(y @ alpha))
        W = np.outer(alpha, alpha) - np.linalg.inv(K)
        return value, ok.kernel_grads(spec, X, W), 0.5 * np.diag(W).copy()


CONFIGS = {
    "default": dict(scale=0.5, linear=True, nonlinear=False, noise=0.1),
    "nonlinear": dict(scale=0.5, linear=True, nonlinear=True, noise=0.1),
    "rq": dict(scale=0.7, linear=True, nonlinear=True, rq=True, noise=0.05),
    "periodic": dict(scale=0.5, per=True, per_period=0.8, linear=True, nonlinear=True, noise=0.1),
    "input_linear": dict(scale=0.5, input_linear=True, linear=False, nonlinear=True, noise=0.1),
    "tied_markov": dict(scale=0.5, scale_tie=True, linear=True, nonlinear=True, markov=1, noise=0.1),
}


def _data(n=40, m=2, p=3, seed=0, missing=0.0, weights=False):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 1, (n, m))
    cols = []
    for i in range(p):
        f = np.sin(4.0 * x @ rng.uniform(0.5, 1.5, m) + i) + (0.5 * cols[-1] if cols else 0.0)
        cols.append(f + 0.1 * rng.standard_normal(n))
    y = np.stack(cols, axis=1)
    if missing:
        y[rng.random(y.shape) < missing] = np.nan
        y[0] = 0.3   # (one complete row)
    w = rng.uniform(0.5, 2.0, y.shape) if weights else None
    return x, y, w


def _layer_objectives(reg, eng, pi, cls):
    """(fast objective, general fg, x0) of layer pi of the conditioned regressor, both over the store reg.vs."""
    x_t, y_t, w_t = eng.tensor(reg.x), eng.tensor(reg.y), eng.tensor(reg.w)
    if y_t is reg.y:
        y_t = y_t.view(y_t.shape)
    y_t._host_nan = torch.isnan(reg.y).numpy()
    y_cached = {k: list(per_output(y_t, w_t, keep=k)) for k in [True, False]}
    gpar = _construct_gpar(reg, reg.vs, reg.m, pi + 1)
    fixed_x, fixed_x_ind = gpar.logpdf(x_t, y_cached, None, only_last_layer=True, outputs=list(range(pi)), return_inputs=True)

    def objective(vs):
        g = _construct_gpar(reg, vs, reg.m, pi + 1)
        return -g.logpdf(fixed_x, y_cached, None, only_last_layer=True, outputs=[pi], x_ind=fixed_x_ind)

    names = [f"{pi}/*"]
    fg, resolved, x0 = objective_and_gradient(objective, reg.vs, names)
    fast = fastfit.build(reg, eng, reg.vs, pi, names, fixed_x, y_cached[bool(reg.impute)][pi], cls=cls)
    assert fast is not None and fast.names == resolved
    return fast, fg, x0


@pytest.mark.parametrize("name", sorted(CONFIGS))
@pytest.mark.parametrize("variant", ["plain", "weights", "missing", "missing_replace"])
def test_prepared_objective_equals_the_autograd_route_on_the_host_side(oracle_engine, name, variant):
    x, y, w = _data(missing=0.2 if variant.startswith("missing") else 0.0, weights=variant == "weights")
    reg = GPARRegressor(normalise_y=variant != "plain", replace=variant == "missing_replace", **CONFIGS[name])
    reg.condition(x, y, w)
    for pi in range(reg.p):
        fast, fg, x0 = _layer_objectives(reg, oracle_engine, pi, NumpyObjective)
        rng = np.random.default_rng(pi)
        for trial in range(3):
            xv = x0 + (0.0 if trial == 0 else 0.3 * rng.standard_normal(x0.shape))
            v_fast, g_fast = fast.fg(xv)
            v_ref, g_ref = fg(xv)
            assert abs(v_fast - v_ref) <= 1e-9 * max(1.0, abs(v_ref)), (name, variant, pi, v_fast, v_ref)
            np.testing.assert_allclose(g_fast, g_ref, rtol=1e-6, atol=1e-7 * max(1.0, np.abs(g_ref).max()))
        reg.vs.set_vector(x0, fast.names)


def test_prepared_objective_leaves_unread_variables_alone_and_reports_failures(oracle_engine):
    x, y, w = _data()
    reg = GPARRegressor(scale=0.5, scale_tie=True, linear=True, nonlinear=True, noise=0.1, normalise_y=False)
    reg.condition(x, y, w)
    fast, fg, x0 = _layer_objectives(reg, oracle_engine, 1, NumpyObjective)
    # layer 1 reads "0/input/scales" (tied) but trains "1/*" only: the tied scales are not among its variables
    assert all(n.startswith("1/") for n in fast.names) and "0/input/scales" in fast.holders

    class Failing(NumpyObjective):
        def _device_eval(self, ck, noise):
            return None

    seen = []
    failing = Failing(fast.eng, fast.vs, fast.names, fast.kernel, fast.noise, fast.holders, fast.X, fast.y, fast.w,
                      general_fg=lambda xv: (seen.append(1), fg(xv))[1])
    v, g = failing.fg(x0)
    v_ref, g_ref = fg(x0)
    assert seen and v == v_ref and np.array_equal(g, g_ref) and failing.fallbacks == 1
    lone = Failing(fast.eng, fast.vs, fast.names, fast.kernel, fast.noise, fast.holders, fast.X, fast.y, fast.w)
    v, g = lone.fg(x0)
    assert np.isnan(v) and not g.any()


def test_fit_takes_the_prepared_objective_where_it_applies_and_trains_the_same_model(oracle_engine, monkeypatch):
    x, y, w = _data(n=30, missing=0.1)
    built = []
    real_build = fastfit.build

    def build(*args, **kwargs):
        obj = real_build(*args, cls=NumpyObjective, **kwargs)
        built.append(obj)
        return obj

    monkeypatch.setattr(fastfit, "build", build)
    kw = dict(scale=0.5, linear=True, nonlinear=True, noise=0.1)
    fast = GPARRegressor(**kw)
    fast.fit(x, y, w, iters=6)
    assert len(built) == 3 and all(b is not None and b.evaluations > 0 for b in built)
    slow = GPARRegressor(**kw)
    slow.fast_fit = False
    slow.fit(x, y, w, iters=6)
    assert len(built) == 3
    a, b = fast.get_variables(), slow.get_variables()
    assert sorted(a) == sorted(b)
    for k in a:
        np.testing.assert_allclose(a[k], b[k], rtol=1e-6, atol=1e-9, err_msg=k)
    # inducing points: the prepared objective does not apply, the general route trains
    sparse = GPARRegressor(x_ind=np.linspace(0, 1, 8)[:, None] * np.ones((1, 2)), **kw)
    before = len(built)
    sparse.fit(x, y, w, iters=2)
    assert all(b is None for b in built[before:])


class _FakeRendezvous(fastfit.LockstepFactor):
    """The protocol of the lock-step rendezvous with its device side replaced by a log (no GPU: threads, a condition, counters)."""

    def _allocate(self, eng):
        self.log = []

    def _stream(self):
        import threading

        return threading.get_ident()

    def _record(self, stream):
        return ("event", stream, len(self.log))

    def _wait(self, stream, event):
        pass

    def _potrf(self, stream, first, count):
        self.log.append((first, count))


def test_lockstep_rendezvous_rounds_are_a_function_of_the_evaluation_counts_only():
    """LockstepFactor: lanes that make different numbers of evaluations (L-BFGS-B line searches differ per layer), train several
    layers one after the other and finish at different times - every round holds exactly the lanes that still have evaluations to
    make, consecutive slots are one batch, nobody deadlocks, and the composition of the rounds is the same whatever the timing."""
    import random
    import threading
    import time

    counts = [[5, 3], [2], [7, 1, 2], [4]]   # evaluations per layer, per lane

    def run(seed):
        rv = _FakeRendezvous(None, 10, len(counts))
        rng = random.Random(seed)
        delays = [[rng.random() * 2e-3 for _ in range(sum(c))] for c in counts]

        def lane(k):
            try:
                it = iter(delays[k])
                for evaluations in counts[k]:
                    for _ in range(evaluations):
                        time.sleep(next(it))
                        rv.factor(k)
            finally:
                rv.leave(k)

        threads = [threading.Thread(target=lane, args=(k,), daemon=True) for k in range(len(counts))]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=20)
            assert not t.is_alive(), "a lane is stuck in the rendezvous"
        return rv.history, rv.log

    totals = [sum(c) for c in counts]
    expected = [tuple(k for k in range(len(counts)) if totals[k] > r) for r in range(max(totals))]
    first = run(0)
    assert first[0] == expected
    assert all(run(seed) == first for seed in (1, 2, 3))
    # runs of consecutive slots are one batch each: round 0 = lanes 0-3 in one call, a round of lanes (0, 2) two calls
    assert first[1][0] == (0, 4) and (0, 1) in first[1] and (2, 1) in first[1]


def test_lockstep_rendezvous_passes_a_failure_to_every_lane_and_a_leaving_lane_completes_the_round():
    import threading

    class Broken(_FakeRendezvous):
        def _potrf(self, stream, first, count):
            raise RuntimeError("launch failed")

    rv = Broken(None, 10, 2)
    seen = []

    def lane(k):
        try:
            rv.factor(k)
        except RuntimeError as exc:
            seen.append((k, str(exc)))
        finally:
            rv.leave(k)

    threads = [threading.Thread(target=lane, args=(k,), daemon=True) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=10)
        assert not t.is_alive()
    assert sorted(seen) == [(0, "launch failed"), (1, "launch failed")]
    # a lane that leaves while the other waits completes the round for it
    rv = _FakeRendezvous(None, 10, 2)
    waiter = threading.Thread(target=lambda: rv.factor(0), daemon=True)
    waiter.start()
    import time

    time.sleep(0.05)
    rv.leave(1)
    waiter.join(timeout=10)
    assert not waiter.is_alive() and rv.history == [(0,)]
    with pytest.raises(RuntimeError):
        rv.factor(1)   # (not a member any more)


# ---- through libgpar_hip.so --------------------------------------------------------------------------------------------------

@pytest.fixture
def hip_engine():
    eng = make_engine("hip")
    previous = set_engine(eng)
    yield eng
    set_engine(previous)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CONFIGS))
@pytest.mark.parametrize("variant,n", [("plain", 100), ("weights", 257), ("missing", 400), ("plain", 1100)])
def test_prepared_objective_on_the_device_same_value_bits_and_gradient(hip_engine, name, variant, n):
    x, y, w = _data(n=n, missing=0.15 if variant == "missing" else 0.0, weights=variant == "weights", seed=n)
    reg = GPARRegressor(normalise_y=False, **CONFIGS[name])
    reg.condition(x, y, w)
    for pi in range(reg.p):
        fast, fg, x0 = _layer_objectives(reg, hip_engine, pi, None)
        rng = np.random.default_rng(pi)
        for trial in range(2):
            xv = x0 + (0.0 if trial == 0 else 0.2 * rng.standard_normal(x0.shape))
            v_fast, g_fast = fast.fg(xv)
            v_ref, g_ref = fg(xv)
            assert v_fast == v_ref, (name, variant, pi, v_fast, v_ref)   # the same launches on the same inputs
            np.testing.assert_allclose(g_fast, g_ref, rtol=1e-11, atol=1e-12 * max(1.0, np.abs(g_ref).max()))
        assert fast.fallbacks == 0
        reg.vs.set_vector(x0, fast.names)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [100, 400, 1024])
def test_fit_trains_the_same_hyperparameters_on_both_routes(hip_engine, n):
    """the same trained hyper-parameters as the general route.  Every evaluation returns the same VALUE to
    the bit and the same gradient up to the summation order of one sum (the noise variance's: n terms added by numpy here, by a
    device reduction there - a relative 1e-16); twenty L-BFGS-B iterations carry that to 1e-9 .. 2e-8 of a trained value."""
    x, y, _ = _data(n=n, m=2, p=4, seed=3)
    kw = dict(scale=0.5, linear=True, nonlinear=False, noise=0.1, normalise_y=False)
    fast = GPARRegressor(**kw)
    fast.fit(x, y, iters=20)
    slow = GPARRegressor(**kw)
    slow.fast_fit = False
    slow.fit(x, y, iters=20)
    a, b = fast.get_variables(), slow.get_variables()
    assert sorted(a) == sorted(b)
    for k in a:
        np.testing.assert_allclose(a[k], b[k], rtol=1e-7, atol=1e-10, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("n,p", [(1100, 4), (1500, 6), (2048, 3)])
def test_lockstep_rendezvous_trains_the_model_of_the_serial_loop(hip_engine, monkeypatch, n, p):
    """fit(fix=True) above ~1000 rows: the lanes' factorisations of a round are ONE gpar_potrf_batch (fastfit.LockstepFactor).  The
    factor of a matrix inside a batch equals the lone factor to rounding (not to the bit), so the trained hyper-parameters are those
    of the serial loop to the optimiser's amplification of rounding; with more layers than lanes a lane trains several layers one
    after the other, and the rendezvous reports as many rounds as the busiest lane made evaluations."""
    from gpar_amd import optimise

    x, y, _ = _data(n=n, m=2, p=p, seed=n)
    kw = dict(scale=0.5, linear=True, nonlinear=True, noise=0.1, normalise_y=False)
    monkeypatch.delenv("GPAR_FIT_THREADS", raising=False)
    monkeypatch.setenv("GPAR_FIT_LOCKSTEP_ROWS", "1000")   # (off by default: see fastfit.lockstep_rows)
    before = optimise.evaluation_count()
    together = GPARRegressor(**kw)
    together.fit(x, y, iters=10)
    evaluations = optimise.evaluation_count() - before
    rounds, batches = together._lockstep_rounds
    assert 0 < rounds <= evaluations and batches >= rounds
    assert rounds < evaluations   # (several lanes per round)
    monkeypatch.setenv("GPAR_FIT_THREADS", "1")
    serial = GPARRegressor(**kw)
    serial.fit(x, y, iters=10)
    a, b = together.get_variables(), serial.get_variables()
    assert sorted(a) == sorted(b)
    for k in a:
        np.testing.assert_allclose(a[k], b[k], rtol=1e-5, atol=1e-8, err_msg=k)
    # off (the default): the lanes factor on their own, bit for bit the serial model
    monkeypatch.delenv("GPAR_FIT_THREADS")
    monkeypatch.delenv("GPAR_FIT_LOCKSTEP_ROWS")
    apart = GPARRegressor(**kw)
    apart.fit(x, y, iters=10)
    assert not hasattr(apart, "_lockstep_rounds")
    for k, v in apart.get_variables().items():
        assert np.array_equal(v, b[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(0, 60, 3))
def test_prepared_objective_on_random_configurations(hip_engine, seed):
    """The fuzz generator of tests/test_fuzz_parity_gpu.py (random kernel families, Markov orders, missing data, weights,
    normalisation, tied scales, 2 to 160 rows): wherever the prepared objective applies, its value is the general route's to the bit and its
    gradient to 1e-9 of the largest component, at the initial point and at a perturbed one.  (This is the test that found the
    one-call-over-the-whole-vector sigmoid: torch's vectorised and scalar exponentials differ in the last bit.)"""
    from .test_fuzz_parity_gpu import _case

    kw, x, y, w, xs = _case(seed)
    if "x_ind" in kw or x.shape[0] > 4096:
        pytest.skip("inducing points / more rows than the prepared objective takes")
    reg = GPARRegressor(**kw)
    reg.condition(x, y, w)
    for pi in range(reg.p):
        fast, fg, x0 = _layer_objectives(reg, hip_engine, pi, None)
        rng = np.random.default_rng(seed * 10 + pi)
        for trial in range(2):
            xv = x0 + (0.0 if trial == 0 else 0.25 * rng.standard_normal(x0.shape))
            v_fast, g_fast = fast.fg(xv)
            v_ref, g_ref = fg(xv)
            if np.isnan(v_ref):
                assert np.isnan(v_fast)
                continue
            assert v_fast == v_ref, (seed, pi, trial, v_fast, v_ref)
            assert np.max(np.abs(g_fast - g_ref)) <= 1e-9 * max(np.abs(g_ref).max(), 1e-3)
        reg.vs.set_vector(x0, fast.names)
