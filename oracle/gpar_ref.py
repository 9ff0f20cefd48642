"""Oracle (TEST INFRASTRUCTURE): a self-contained numpy GPAR log marginal likelihood and posterior-mean chain.

Restates the orchestration of /root/reference/gpar/model.py:178-243 (`GPAR.logpdf`), :279-322 (`_obs`,
`_update_inputs`), :325-362 (`per_output`) and the per-layer kernel of /root/reference/gpar/regression.py:92-180
directly from a `{name: value}` dictionary of hyper-parameters (the dictionary `GPARRegressor.get_variables()`
returns), without importing anything from the product.  Dense and inducing-point (VFE) paths.
"""
import numpy as np

from . import gp_ref

__all__ = ["layer_spec", "gpar_logpdf"]


def _indices(m, pi, markov):
    p_last = pi - 1
    p_start = 0 if markov is None else max(p_last - (markov - 1), 0)
    return list(range(m)), list(range(m + p_start, m + p_last + 1))


def layer_spec(hypers, m, pi, config):
    """Kernel dict (oracle/kernels.py format) and noise variance of layer `pi` (regression.py:92-180)."""
    g = lambda name: np.asarray(hypers[name], dtype=np.float64)
    m_inds, p_inds = _indices(m, pi, config.get("markov"))
    rq = config.get("rq", False)
    terms = []

    def nonlin(prefix, cols, scales):
        f = {"type": "rq" if rq else "eq", "cols": cols, "scales": list(np.atleast_1d(scales)), "periods": None, "alpha": 0.0}
        if rq:
            f["alpha"] = float(g(f"{prefix}/alpha"))
        return f

    tie = 0 if config.get("scale_tie", False) else pi
    terms.append({"coef": float(g(f"{pi}/input/var")), "factors": [nonlin(f"{pi}/input", m_inds, g(f"{tie}/input/scales"))]})
    if config.get("per", False):
        terms.append(
            {
                "coef": float(g(f"{pi}/input/per/var")),
                "factors": [
                    {"type": "eq", "cols": m_inds, "scales": list(g(f"{pi}/input/per/scales")),
                     "periods": list(np.atleast_1d(g(f"{pi}/input/per/pers"))), "alpha": 0.0},
                    {"type": "eq", "cols": m_inds, "scales": list(np.atleast_1d(g(f"{pi}/input/per/decay"))), "periods": None, "alpha": 0.0},
                ],
            }
        )
    if config.get("input_linear", False):
        terms.append({"coef": 1.0, "factors": [{"type": "linear", "cols": m_inds, "scales": list(np.atleast_1d(g(f"{pi}/input/lin/scales"))), "periods": None, "alpha": 0.0}]})
        terms.append({"coef": float(g(f"{pi}/input/lin/const")), "factors": []})
    if config.get("linear", True) and pi > 0:
        terms.append({"coef": 1.0, "factors": [{"type": "linear", "cols": p_inds, "scales": list(np.atleast_1d(g(f"{pi}/output/lin/scales"))), "periods": None, "alpha": 0.0}]})
    if config.get("nonlinear", False) and pi > 0:
        terms.append({"coef": float(g(f"{pi}/output/nonlin/var")), "factors": [nonlin(f"{pi}/output/nonlin", p_inds, g(f"{pi}/output/nonlin/scales"))]})
    return {"terms": terms}, float(g(f"{pi}/noise"))


def gpar_logpdf(x, y, w, hypers, config, impute=False, replace=False, eps=1e-12, x_ind=None, sample_missing=False, seed=0):
    """Sum over layers of log N(y_i; 0, K_i([x, y_<i]) + noise_i / w_i) - with `x_ind`, of the VFE bounds - with the
    reference's missing-data rules (model.py:178-243, 279-322): rows kept per layer (`per_output`), observations filtered
    (`_obs`), the next input column = observed values, imputed / replaced by posterior means (`_update_inputs`), and the
    inducing inputs extended by the posterior mean at the inducing inputs.

    `sample_missing` (model.py:229-237): before the inputs are updated, the missing entries of the column are drawn from
    the layer's posterior at their inputs, `mean + chol(cov + diag(noise / w) + eps I) z`; the normals z come from the
    counter-based stream the product uses (oracle/philox.py: call k of a computation uses offset k), so that a product run
    with the same seed draws the same imputations and the two values can be compared to rounding."""
    from . import philox

    x = np.asarray(x, dtype=np.float64)
    x = x[:, None] if x.ndim == 1 else x
    y = np.asarray(y, dtype=np.float64)
    w = np.ones_like(y) if w is None else np.asarray(w, dtype=np.float64)
    sparse = x_ind is not None
    if sparse:
        x_ind = np.asarray(x_ind, dtype=np.float64)
        x_ind = x_ind[:, None] if x_ind.ndim == 1 else x_ind
    m, p = x.shape[1], y.shape[1]
    available = ~np.isnan(y)
    total = 0.0
    calls = 0
    for i in range(p):
        mask = available[:, i].copy()
        if (impute or sample_missing) and i < p - 1:
            mask |= available[:, i + 1 :].any(axis=1)
        x, yi, wi = x[mask], y[mask, i], w[mask, i]
        y, w, available = y[mask], w[mask], available[mask]
        spec, noise = layer_spec(hypers, m, i, config)
        have = ~np.isnan(yi)
        if sparse:
            total += gp_ref.vfe_bound(spec, x[have], yi[have], noise / wi[have], x_ind, eps=eps)
        else:
            total += gp_ref.logpdf(spec, x[have], yi[have], noise / wi[have], eps=eps)
        if i < p - 1:

            def estimate(points):
                if sparse:
                    return gp_ref.vfe_posterior(spec, x[have], yi[have], noise / wi[have], x_ind, points, eps=eps)[0]
                return gp_ref.posterior(spec, x[have], yi[have], noise / wi[have], points, eps=eps)[0]

            col = yi.copy()
            seen = have  # `available` as `_update_inputs` sees it: after the draws below, nothing is missing any more
            if sample_missing and (~have).any():
                if sparse:
                    mean, cov = gp_ref.vfe_posterior(spec, x[have], yi[have], noise / wi[have], x_ind, x[~have], eps=eps)
                else:
                    mean, cov = gp_ref.posterior(spec, x[have], yi[have], noise / wi[have], x[~have], eps=eps)
                S = cov + np.diag(noise / wi[~have] + eps)
                z = philox.randn(seed, calls, int((~have).sum()), 1)[:, 0]
                calls += 1
                col[~have] = mean + np.linalg.cholesky(S) @ z
                seen = np.ones_like(have)
            if (impute and (~seen).any()) or (replace and seen.any()):
                mean = estimate(x)
                if impute:
                    col[~seen] = mean[~seen]
                if replace:
                    col[seen] = mean[seen]
            if sparse:
                x_ind = np.concatenate([x_ind, estimate(x_ind)[:, None]], axis=1)
            x = np.concatenate([x, col[:, None]], axis=1)
    return total
