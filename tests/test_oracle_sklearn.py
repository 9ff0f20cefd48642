"""Pins the CPU oracle against an INDEPENDENT published implementation: scikit-learn's GaussianProcessRegressor.

The reference's own numerical stack (stheno / mlkernels / lab) cannot be installed here, so the oracle cannot be
checked against the reference itself ("parity unpinned", oracle/__init__.py).  scikit-learn implements the same
published definitions the reference's kernels follow (Rasmussen & Williams 2006, ch. 2 and 4):

    mlkernels EQ().stretch(s)            = exp(-r^2 / (2 s^2))                    = sklearn RBF(length_scale=s)
    mlkernels RQ(alpha).stretch(s)       = (1 + r^2 / (2 alpha s^2))^-alpha       = sklearn RationalQuadratic(s, alpha)
    mlkernels Linear()                   = <x, x'>                                = sklearn DotProduct(sigma_0=0)
    mlkernels EQ().stretch(s).periodic(p) = exp(-2 sin^2(pi d / p) / s^2)         = sklearn ExpSineSquared(s, p)

so agreement of Gram matrices, log marginal likelihoods and posterior moments to ~1e-10 removes "the oracle and the
product share a misreading of the formulas" as a failure mode.  Sums, products and constant scalings are exercised
because GPAR's layer kernels are sums of products (gpar/regression.py:92-180 in the reference).
"""
import numpy as np
import pytest

from oracle import gp_ref
from oracle import kernels as ok

sk_gp = pytest.importorskip("sklearn.gaussian_process")
from sklearn.gaussian_process.kernels import RBF, ConstantKernel, DotProduct, ExpSineSquared, RationalQuadratic  # noqa: E402


def _f(type, cols, scales, periods=None, alpha=0.0):
    return {"type": type, "cols": cols, "scales": scales, "periods": periods, "alpha": alpha}


def _cases():
    sum_spec = {
        "terms": [
            {"coef": 1.7, "factors": [_f("eq", [0, 1], [0.6, 0.6])]},
            {"coef": 0.8, "factors": [_f("rq", [0, 1], [0.9, 0.9], alpha=1.3)]},
            {"coef": 0.5, "factors": [_f("linear", [0, 1], [1.0, 1.0])]},
        ]
    }
    sum_sk = ConstantKernel(1.7) * RBF(0.6) + ConstantKernel(0.8) * RationalQuadratic(0.9, 1.3) + ConstantKernel(0.5) * DotProduct(0.0)
    prod_spec = {
        "terms": [
            {"coef": 1.2, "factors": [_f("eq", [0], [1.1, 1.1], periods=[0.7]), _f("rq", [0], [0.5], alpha=2.0)]},
            {"coef": 0.3, "factors": [_f("eq", [0], [0.25])]},
        ]
    }
    prod_sk = ConstantKernel(1.2) * ExpSineSquared(1.1, 0.7) * RationalQuadratic(0.5, 2.0) + ConstantKernel(0.3) * RBF(0.25)
    return [("sum_eq_rq_linear", sum_spec, sum_sk, 2), ("periodic_times_rq_plus_eq", prod_spec, prod_sk, 1)]


@pytest.mark.parametrize("name,spec,sk_kernel,m", _cases(), ids=[c[0] for c in _cases()])
def test_oracle_matches_scikit_learn(name, spec, sk_kernel, m):
    rng = np.random.default_rng(11)
    n, ns, noise = 60, 17, 0.05
    x = rng.uniform(-1, 1, (n, m))
    xs = rng.uniform(-1, 1, (ns, m))
    y = np.sin(3 * x[:, 0]) + 0.1 * rng.standard_normal(n)

    np.testing.assert_allclose(ok.gram(spec, x), sk_kernel(x), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(ok.gram(spec, xs, x), sk_kernel(xs, x), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(ok.gram_diag(spec, x), sk_kernel.diag(x), rtol=1e-12)

    gpr = sk_gp.GaussianProcessRegressor(kernel=sk_kernel, alpha=noise, optimizer=None, normalize_y=False).fit(x, y)
    # log marginal likelihood (R&W eq. 2.30), posterior mean / covariance of the latent function (eq. 2.23-2.24)
    assert gp_ref.logpdf(spec, x, y, noise, eps=0.0) == pytest.approx(gpr.log_marginal_likelihood_value_, rel=1e-10)
    mean, cov = gp_ref.posterior(spec, x, y, noise, xs, eps=0.0)
    sk_mean, sk_cov = gpr.predict(xs, return_cov=True)
    np.testing.assert_allclose(mean, sk_mean, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(cov, sk_cov, rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("name,spec,sk_kernel,m", _cases(), ids=[c[0] for c in _cases()])
def test_oracle_logpdf_matches_scipy_multivariate_normal(name, spec, sk_kernel, m):
    """A third route to the dense log marginal likelihood that shares nothing with a Cholesky factorisation:
    scipy.stats.multivariate_normal works from the eigendecomposition of the covariance (pseudo-determinant and
    pseudo-inverse).  Agreement to 1e-9 at a conditioning of ~1e4."""
    from scipy.stats import multivariate_normal

    rng = np.random.default_rng(12)
    n, noise = 80, 0.05
    x = rng.uniform(-1, 1, (n, m))
    y = np.cos(2 * x[:, 0]) + 0.1 * rng.standard_normal(n)
    K = ok.gram(spec, x) + noise * np.eye(n)
    ref = multivariate_normal(mean=np.zeros(n), cov=K, allow_singular=False).logpdf(y)
    assert gp_ref.logpdf(spec, x, y, noise, eps=0.0) == pytest.approx(ref, rel=1e-9)
