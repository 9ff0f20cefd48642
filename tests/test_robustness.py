"""Behaviour around the numerical core that the advisor's review asked to pin: a compiled kernel specification follows in-place
updates of its parameter tensors, a hand-off timeout on an inference path is retried in safe mode (and is NOT retried inside a
training objective), FITC is a valid `sparse_method`, an unknown one is refused at construction, and importing the package
leaves the process environment alone."""
import contextlib
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_compiled_specification_follows_in_place_parameter_updates():
    from gpar_amd.engine import HipEngine
    from gpar_amd.kernels import EQ, Linear

    scales = torch.tensor([0.5, 2.0], dtype=torch.float64)
    var = torch.tensor(1.5, dtype=torch.float64)
    np_scales = np.array([3.0, 4.0])
    k = var * EQ().stretch(scales) + Linear().stretch(np_scales)
    compile_ = lambda: HipEngine.compile(None, k, 2)  # the method keeps its cache on the kernel object
    ck = compile_()
    assert compile_() is ck  # unchanged parameters: the cached specification
    assert ck.fspec.inv_scale[0] == 2.0 and ck.kspec.coef[0] == 1.5
    scales.fill_(0.25)  # what an optimiser step does to a leaf
    ck2 = compile_()
    assert ck2 is not ck and ck2.fspec.inv_scale[0] == 4.0 and ck2.fspec.inv_scale[1] == 4.0
    with torch.no_grad():
        var.mul_(2.0)
    assert compile_().kspec.coef[0] == 3.0
    np_scales[0] = 6.0  # numpy parameters are fingerprinted by value
    assert compile_().fspec.inv_scale[2] == 1.0 / 6.0
    assert compile_() is compile_()


class _FlakyEngine:
    """Oracle engine + the HIP engine's safe-mode switch; `fail` evaluations raise a hand-off timeout first."""

    def __new__(cls, fail):
        from oracle.engine import OracleEngine

        class Flaky(OracleEngine):
            def __init__(self):
                super().__init__(seed=3)
                self.fail, self.safe_entries, self.in_safe = fail, 0, False

            @contextlib.contextmanager
            def safe_mode(self):
                self.safe_entries += 1
                self.in_safe = True
                try:
                    yield
                finally:
                    self.in_safe = False

            def potrf_(self, A, nf=None):
                from gpar_amd.engine import HandOffTimeoutError

                if self.fail > 0 and not self.in_safe:
                    self.fail -= 1
                    raise HandOffTimeoutError(-77)
                return super().potrf_(A, nf=nf)

        return Flaky()


def _tiny():
    rng = np.random.default_rng(2)
    x = rng.uniform(0, 1, (30, 1))
    y = np.stack([np.sin(5 * x[:, 0]), np.cos(4 * x[:, 0])], axis=1) + 0.05 * rng.standard_normal((30, 2))
    return x, y


def test_hand_off_timeout_on_an_inference_path_is_retried_in_safe_mode():
    from gpar_amd.engine import set_engine
    from gpar_amd.regression import GPARRegressor

    x, y = _tiny()
    assert torch.is_grad_enabled()  # the default mode of every inference call: the retry must not depend on it
    clean = set_engine(_FlakyEngine(0))
    try:
        want = float(GPARRegressor(noise=0.1).logpdf(x, y))
        eng = _FlakyEngine(1)
        set_engine(eng)
        reg = GPARRegressor(noise=0.1)
        assert float(reg.logpdf(x, y)) == want and eng.safe_entries == 1
        eng.fail = 1
        reg.condition(x, y)
        sample = reg.sample(x[:5], posterior=True)  # conditioning (`gpar | data`) goes through the same retry
        assert eng.safe_entries == 2 and np.all(np.isfinite(sample))
    finally:
        set_engine(clean)


def test_hand_off_timeout_inside_a_training_objective_is_not_retried():
    from gpar_amd.engine import HandOffTimeoutError, set_engine
    from gpar_amd.regression import GPARRegressor

    x, y = _tiny()
    eng = _FlakyEngine(0)
    previous = set_engine(eng)
    try:
        reg = GPARRegressor(noise=0.1)
        with torch.no_grad():
            reg.logpdf(x, y)  # instantiates the variables
        reg.vs.requires_grad(True)
        eng.fail = 1
        with pytest.raises(HandOffTimeoutError):
            reg.logpdf(torch.tensor(x), torch.tensor(y))
        assert eng.safe_entries == 0
        reg.vs.requires_grad(False)
        eng.fail = 1
        assert np.isfinite(float(reg.logpdf(x, y))) and eng.safe_entries == 1
    finally:
        set_engine(previous)


def test_non_positive_pivot_gets_a_second_opinion_on_the_unfused_path(monkeypatch):
    """A matrix at the edge of numerical definiteness (K_zz + 1e-12 of many inducing inputs on one axis): the fused panel path and
    LAPACK's arithmetic do not fail on the same ones, so a non-positive pivot is confirmed on the unfused path before it is
    reported - under autograd too (the evaluation builds its graph afresh) - and reported if that path repeats it."""
    from gpar_amd.engine import NotPositiveDefiniteError, set_engine
    from gpar_amd.regression import GPARRegressor

    from oracle.engine import OracleEngine

    class Marginal(type(_FlakyEngine(0))):
        always = False

        def potrf_(self, A, nf=None):
            if self.always or (self.fail > 0 and not self.in_safe):
                self.fail -= 1
                raise NotPositiveDefiniteError(148)
            return OracleEngine.potrf_(self, A, nf=nf)

    x, y = _tiny()
    eng = Marginal()
    previous = set_engine(eng)
    try:
        reg = GPARRegressor(noise=0.1)
        want = float(reg.logpdf(x, y))
        eng.fail = 1
        assert float(reg.logpdf(x, y)) == want and eng.safe_entries == 1
        reg.vs.requires_grad(True)
        eng.fail = 1
        value = reg.logpdf(torch.tensor(x), torch.tensor(y))
        value.backward()
        assert float(value.detach()) == want and eng.safe_entries == 2
        reg.vs.requires_grad(False)
        eng.always = True  # the unfused path agrees: the error is the caller's
        with pytest.raises(NotPositiveDefiniteError):
            reg.logpdf(x, y)
        assert eng.safe_entries == 3
        eng.always, eng.fail = False, 1
        monkeypatch.setenv("GPAR_NOTPD_RETRY", "0")
        with pytest.raises(NotPositiveDefiniteError):
            reg.logpdf(x, y)
        assert eng.safe_entries == 3
    finally:
        set_engine(previous)


def test_sparse_method_is_validated_at_construction():
    from gpar_amd.regression import GPARRegressor

    with pytest.raises(ValueError, match="sparse_method"):
        GPARRegressor(sparse_method="ftic")
    for method in ("vfe", "fitc", "dtc"):
        assert GPARRegressor(sparse_method=method).sparse_method == method


def test_importing_the_package_leaves_the_environment_alone():
    code = ("import os; before = dict(os.environ); import gpar_amd; import gpar_amd.engine; "
            "assert dict(os.environ) == before, set(os.environ) ^ set(before); print('ok')")
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-1500:]


def test_condition_does_not_touch_the_global_thread_count(monkeypatch):
    from gpar_amd.regression import GPARRegressor

    calls = []
    monkeypatch.setattr(torch, "set_num_threads", lambda k: calls.append(k))
    x, y = _tiny()
    reg = GPARRegressor()
    reg.condition(x, y)
    assert calls == []
    np.testing.assert_allclose(reg.y.numpy().mean(0), 0.0, atol=1e-12)
    np.testing.assert_allclose(reg.y.numpy().std(0), 1.0, rtol=1e-12)
    assert reg.w.shape == reg.y.shape and bool((reg.w == 1).all())
