"""Observations of a posterior process: `Obs` / `PseudoObs` built on `f | obs`, the log-density under a sparse posterior
and conditioning twice, for every combination of exact and inducing-point observations - the paths
`GPARRegressor.logpdf(..., posterior=True)` takes with `x_ind` set (reference gpar/regression.py:493-499 -> stheno's
measure algebra).  Checked against dense closed forms that share no code with the product (oracle/gp_ref.Process);
runs on the CPU oracle engine here and, unchanged, through the HIP library on the GPU."""
import numpy as np
import pytest

from gpar_amd.gp import GP, Obs, PseudoObs
from gpar_amd.kernels import EQ, Linear
from gpar_amd.regression import GPARRegressor
from oracle import gp_ref
from oracle import kernels as ok


def _setup(seed=0):
    rng = np.random.default_rng(seed)
    kernel = 1.3 * EQ().stretch(np.array([0.6, 0.9])) + Linear().stretch(np.array([3.0, 4.0])) + 0.2
    spec = ok.spec_to_dict(kernel.resolve(2))
    pts = lambda n: rng.uniform(0, 1, (n, 2))
    data = dict(x1=pts(23), x2=pts(17), xs=pts(11), z1=pts(7), z2=pts(6))
    data["y1"] = np.sin(3 * data["x1"][:, 0]) + 0.1 * rng.standard_normal(23)
    data["y2"] = np.cos(2 * data["x2"][:, 1]) + 0.1 * rng.standard_normal(17)
    data["ys"] = rng.standard_normal(11)
    data["d1"] = rng.uniform(0.05, 0.2, 23)
    data["d2"] = rng.uniform(0.05, 0.2, 17)
    data["ds"] = rng.uniform(0.05, 0.2, 11)
    return kernel, spec, data


def _condition(f, ref, kind, x, y, d, z):
    """(product posterior, reference posterior) after one conditioning step of the given kind."""
    if kind == "dense":
        return f | Obs(f(x, d), y), ref.condition(x, y, d)
    return f | PseudoObs(f(z), f(x, d), y), ref.condition_sparse(x, y, d, z)


@pytest.mark.parametrize("first", ["dense", "sparse"])
@pytest.mark.parametrize("second", ["dense", "sparse"])
def test_observations_of_a_posterior(engine, first, second):
    kernel, spec, D = _setup()
    f = GP(kernel)
    ref = gp_ref.Process.prior(spec)
    f1, r1 = _condition(f, ref, first, D["x1"], D["y1"], D["d1"], D["z1"])

    # log-density / bound of NEW observations under the posterior
    if second == "dense":
        got = float(f1.measure.logpdf(Obs(f1(D["x2"], D["d2"]), D["y2"])))
        want = r1.logpdf(D["x2"], D["y2"], D["d2"])
    else:
        got = float(f1.measure.logpdf(PseudoObs(f1(D["z2"]), f1(D["x2"], D["d2"]), D["y2"])))
        want = r1.vfe_bound(D["x2"], D["y2"], D["d2"], D["z2"])
    assert got == pytest.approx(want, rel=1e-9, abs=1e-9)

    # conditioning twice, then moments and a further log-density
    f2, r2 = _condition(f1, r1, second, D["x2"], D["y2"], D["d2"], D["z2"])
    np.testing.assert_allclose(f2.mean(D["xs"]).cpu().numpy().reshape(-1), r2.mean(D["xs"]), rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(f2(D["xs"]).var().cpu().numpy(), r2.k(D["xs"], D["xs"]), rtol=1e-7, atol=1e-9)
    mean, var = f2(D["xs"], D["ds"]).marginals()
    np.testing.assert_allclose(mean.cpu().numpy(), r2.mean(D["xs"]), rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(var.cpu().numpy(), np.diag(r2.k(D["xs"], D["xs"])) + D["ds"], rtol=1e-7, atol=1e-9)
    assert float(f2(D["xs"], D["ds"]).logpdf(D["ys"])) == pytest.approx(r2.logpdf(D["xs"], D["ys"], D["ds"]), rel=1e-8)


def test_observations_must_belong_to_the_conditioned_process(engine):
    kernel, _, D = _setup()
    f = GP(kernel)
    post = f | Obs(f(D["x1"], D["d1"]), D["y1"])
    with pytest.raises(ValueError):
        post | Obs(f(D["x2"], D["d2"]), D["y2"])  # observations of the prior handed to the posterior
    with pytest.raises(ValueError):
        PseudoObs(f(D["z1"]), post(D["x2"], D["d2"]), D["y2"])  # inducing points of one process, data of another


@pytest.mark.parametrize("kw", [dict(), dict(replace=True), dict(impute=True, replace=True)])
def test_sparse_regressor_posterior_logpdf(engine, kw):
    """`GPARRegressor(x_ind=...).condition(...); reg.logpdf(x, y, posterior=True)` used to raise."""
    rng = np.random.default_rng(3)
    x = np.linspace(0, 1, 30)
    y = np.stack([np.sin(5 * x), np.cos(4 * x) * x, x**2], axis=1) + 0.05 * rng.standard_normal((30, 3))
    x_new = rng.uniform(0, 1, 12)
    y_new = np.stack([np.sin(5 * x_new), np.cos(4 * x_new) * x_new, x_new**2], axis=1)
    reg = GPARRegressor(x_ind=np.linspace(0, 1, 8), scale=0.3, linear=True, nonlinear=True, noise=0.05, normalise_y=False, **kw)
    reg.condition(x, y)
    value = float(reg.logpdf(x_new, y_new, posterior=True))
    assert np.isfinite(value)
    # the posterior explains held-out data from the same functions far better than the prior does
    assert value > float(reg.logpdf(x_new, y_new)) + 10.0
    # layer 0 in closed form: VFE bound of the new data under the sparse posterior of the old data
    f, noise = reg_layer0(reg)
    spec = ok.spec_to_dict(f.kernel.resolve(1))
    ref = gp_ref.Process.prior(spec).condition_sparse(x[:, None], y[:, 0], float(noise), np.linspace(0, 1, 8)[:, None])
    want0 = ref.vfe_bound(x_new[:, None], y_new[:, 0], float(noise), np.linspace(0, 1, 8)[:, None])
    got0 = float(reg_posterior_layer_logpdf(reg, x_new, y_new))
    assert got0 == pytest.approx(want0, rel=1e-8)


def reg_layer0(reg):
    from gpar_amd.regression import _construct_gpar

    return _construct_gpar(reg, reg.vs, reg.m, 1).layers[0]()


def reg_posterior_layer_logpdf(reg, x_new, y_new):
    from gpar_amd.regression import _construct_gpar

    gpar = _construct_gpar(reg, reg.vs, reg.m, 1) | (reg.x, reg.y[:, :1], reg.w[:, :1])
    return gpar.logpdf(x_new, y_new[:, :1], np.ones((len(x_new), 1)))


@pytest.mark.parametrize("method", ["vfe", "dtc", "fitc"])
def test_inducing_point_approximations(engine, method):
    """stheno's PseudoObsVFE / PseudoObsDTC / PseudoObsFITC: bound / log-density and posterior moments against the dense
    closed forms; the analytic gradient of the VFE, DTC and FITC objectives against central differences."""
    import torch

    from gpar_amd.gp import PseudoObs, PseudoObsDTC, PseudoObsFITC
    from gpar_amd.regression import GPARRegressor

    cls = {"vfe": PseudoObs, "dtc": PseudoObsDTC, "fitc": PseudoObsFITC}[method]
    kernel, spec, D = _setup(seed=2)
    f = GP(kernel)
    ref = gp_ref.Process.prior(spec)
    obs = cls(f(D["z1"]), f(D["x1"], D["d1"]), D["y1"])
    want = ref.vfe_bound(D["x1"], D["y1"], D["d1"], D["z1"], method=method)
    assert float(obs.logpdf()) == pytest.approx(want, rel=1e-9)
    post, rpost = f | obs, ref.condition_sparse(D["x1"], D["y1"], D["d1"], D["z1"], method=method)
    np.testing.assert_allclose(post.mean(D["xs"]).cpu().numpy().reshape(-1), rpost.mean(D["xs"]), rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(post(D["xs"]).var().cpu().numpy(), rpost.k(D["xs"], D["xs"]), rtol=1e-7, atol=1e-9)

    # through the regressor, and its training gradient (one layer: no inputs forwarded between layers)
    rng = np.random.default_rng(5)
    x = rng.uniform(0, 1, (40, 1))
    y = np.sin(5 * x) + 0.05 * rng.standard_normal((40, 1))
    reg = GPARRegressor(x_ind=np.linspace(0, 1, 7), scale=0.3, linear=True, nonlinear=True, noise=0.05, normalise_y=False, sparse_method=method)
    base = float(reg.logpdf(x, y))
    assert np.isfinite(base)
    reg.vs.requires_grad(True)
    reg.logpdf(torch.tensor(x), torch.tensor(y)).backward()
    latents = reg.vs.get_vars()
    grad = np.concatenate([(v.grad if v.grad is not None else torch.zeros_like(v)).numpy().reshape(-1) for v in latents])
    reg.vs.requires_grad(False)
    names = reg.vs.names
    x0 = reg.vs.get_vector(names)
    fd = np.zeros_like(x0)
    for i in range(len(x0)):
        for sgn in (+1, -1):
            xi = x0.copy()
            xi[i] += sgn * 1e-5
            reg.vs.set_vector(xi, names)
            fd[i] += sgn * float(reg.logpdf(x, y)) / 2e-5
    reg.vs.set_vector(x0, names)
    np.testing.assert_allclose(grad, fd, rtol=2e-5, atol=1e-6 * np.max(np.abs(fd)))


def test_stacked_input_sets_behave_like_the_list_of_their_blocks():
    """gp.Stacked (the per-sample design matrices of ancestral sampling, kept in one matrix from layer to layer): length,
    indexing, contiguous slices, iteration and appended columns agree with the list-of-matrices form it replaces."""
    import torch

    from gpar_amd.gp import Stacked

    g = torch.Generator().manual_seed(3)
    x = torch.rand(7, 2, generator=g, dtype=torch.float64)
    S = 4
    st = Stacked.repeat(x, S)
    assert len(st) == S and st.rows == 7 and all(torch.equal(b, x) for b in st)
    fed = torch.rand(7, S, generator=g, dtype=torch.float64)
    st2 = st.with_columns(fed)
    as_list = [torch.cat([x, fed[:, s : s + 1]], dim=1) for s in range(S)]
    assert st2.matrix.shape == (7 * S, 3)
    for s in range(S):
        assert torch.equal(st2[s], as_list[s])
    assert torch.equal(st2[-1], as_list[-1])
    part = st2[1:3]
    assert len(part) == 2 and torch.equal(part[0], as_list[1]) and torch.equal(part[1], as_list[2])
    means = torch.rand(7 * S, 1, generator=g, dtype=torch.float64)   # already stacked: one column per set, set-major
    st3 = st2.with_columns(means)
    for s in range(S):
        assert torch.equal(st3[s], torch.cat([as_list[s], means[7 * s : 7 * (s + 1)]], dim=1))
    one = Stacked.repeat(x, 1).with_columns(fed[:, :1])
    assert len(one) == 1 and torch.equal(one[0], as_list[0])
    try:
        st2[::2]
    except IndexError:
        pass
    else:
        raise AssertionError("strided slices must be rejected")
