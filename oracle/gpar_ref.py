"""Oracle (TEST INFRASTRUCTURE): a self-contained numpy GPAR - log marginal likelihood, conditioning chain, predictive moments,
finite-difference gradients.

Restates the orchestration of /root/reference/gpar/model.py:178-243 (`GPAR.logpdf`), :279-322 (`_obs`,
`_update_inputs`), :325-362 (`per_output`), :116-149 (`GPAR.__or__`: conditioning), :245-277 (`GPAR.sample`) and the per-layer
kernel of /root/reference/gpar/regression.py:92-180 directly from a `{name: value}` dictionary of hyper-parameters (the
dictionary `GPARRegressor.get_variables()` returns), without importing anything from the product: own kernels, own
missing-data bookkeeping, `slogdet` + `solve` (oracle/gp_ref.py) instead of any factorisation the product composes.  Dense and
inducing-point (VFE) paths.

What each public function pins:
  gpar_logpdf            GPARRegressor.logpdf / GPAR.logpdf (values);
  gpar_predict_moments   condition + predict when `replace=True` (regression.py:339-389, 566-597; model.py:116-149, 245-277, 291-322):
                         the posterior mean replaces every sampled column, so layer i's inputs at x* are deterministic and every
                         sample of layer i is a draw from N(mean_i(x*_i), cov_i(x*_i) [+ noise_i / w*]) - predictive means and
                         variances in closed form, which the Monte-Carlo `predict` converges to and the product's analytic
                         `predict_moments` must equal;
  fd_gradient            the gradient `fit` needs (regression.py:434-459, autograd there): central differences of gpar_logpdf in
                         every entry of the hyper-parameter dictionary, with `bounds` / `to_unconstrained` (varz's `bnd` map,
                         regression.py:101-173) to carry it to the optimiser's variables.
"""
import numpy as np

from . import gp_ref
from . import kernels as oracle_kernels

__all__ = ["layer_spec", "gpar_logpdf", "gpar_predict_moments", "gpar_sample", "fd_gradient", "bounds", "to_unconstrained"]


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


def _normalisation(y, normalise_y):
    """Per-output mean and POPULATION standard deviation over the observed entries (regression.py:339-389; lab's B.std is ddof = 0;
    an output with zero spread keeps scale 1, :366-370)."""
    p = y.shape[1]
    if not normalise_y:
        return np.zeros(p), np.ones(p)
    mean = np.array([np.mean(y[~np.isnan(y[:, i]), i]) for i in range(p)])
    std = np.array([np.std(y[~np.isnan(y[:, i]), i]) for i in range(p)])
    std = np.where(std > 0, std, 1.0)
    return mean, std


def gpar_predict_moments(x, y, w, hypers, config, xs, ws=None, latent=False, impute=True, replace=True, eps=1e-12, x_ind=None,
                         normalise_y=True):
    """Predictive mean and variance (n* x p each) of a GPAR conditioned on (x, y, w), at inputs xs, for `replace=True`.

    Conditioning (model.py:116-149 through `_update_inputs`, :291-322): layer i sees the rows `per_output` keeps, is conditioned
    on its observed entries, and hands the next layer the column in which observed entries are REPLACED by the posterior mean and
    missing ones imputed by it (impute and replace: the whole column is the posterior mean, :310-311).  Prediction (model.py:245-277
    with obs = None: `estimate` is the mean of the conditioned layer): x*_{i+1} = [x*_i, mean_i(x*_i)], a deterministic chain, so
    the draws of layer i are N(mean_i(x*_i), cov_i(x*_i)) + (unless `latent`) N(0, noise_i / w*_i) and the Monte-Carlo mean /
    variance of `predict` have these limits.  Outputs are un-normalised as regression.py:551-552 does."""
    if not replace:
        raise ValueError("closed-form predictive moments exist only for replace=True (otherwise samples are fed forward)")
    x = np.asarray(x, dtype=np.float64)
    x = x[:, None] if x.ndim == 1 else x
    xs = np.asarray(xs, dtype=np.float64)
    xs = xs[:, None] if xs.ndim == 1 else xs
    y = np.asarray(y, dtype=np.float64)
    y = y[:, None] if y.ndim == 1 else y
    w = np.ones_like(y) if w is None else np.asarray(w, dtype=np.float64)
    m, p = x.shape[1], y.shape[1]
    ws = np.ones((xs.shape[0], p)) if ws is None else np.asarray(ws, dtype=np.float64)
    offset, scale = _normalisation(y, normalise_y)
    y = (y - offset) / scale
    sparse = x_ind is not None
    if sparse:
        x_ind = np.asarray(x_ind, dtype=np.float64)
        x_ind = x_ind[:, None] if x_ind.ndim == 1 else x_ind
    available = ~np.isnan(y)
    means, variances = [], []
    for i in range(p):
        mask = available[:, i].copy()
        if impute and i < p - 1:
            mask |= available[:, i + 1:].any(axis=1)
        x, yi, wi = x[mask], y[mask, i], w[mask, i]
        y, w, available = y[mask], w[mask], available[mask]
        spec, noise = layer_spec(hypers, m, i, config)
        have = ~np.isnan(yi)

        def posterior(points):
            if sparse:
                return gp_ref.vfe_posterior(spec, x[have], yi[have], noise / wi[have], x_ind, points, eps=eps)
            return gp_ref.posterior(spec, x[have], yi[have], noise / wi[have], points, eps=eps)

        mean_s, cov_s = posterior(xs)
        var_s = np.diag(cov_s) + (0.0 if latent else noise / ws[:, i])
        means.append(mean_s * scale[i] + offset[i])
        variances.append(var_s * scale[i] ** 2)
        if i < p - 1:
            col = yi.copy()
            mean_x = posterior(x)[0]
            if impute:
                col[~have] = mean_x[~have]
            col[have] = mean_x[have]   # replace
            if sparse:
                x_ind = np.concatenate([x_ind, posterior(x_ind)[0][:, None]], axis=1)
            x = np.concatenate([x, col[:, None]], axis=1)
            xs = np.concatenate([xs, mean_s[:, None]], axis=1)
    return np.stack(means, axis=1), np.stack(variances, axis=1)


def gpar_sample(x, y, w, hypers, config, xs, ws=None, num_samples=1, latent=False, impute=False, replace=False, eps=1e-12,
                normalise_y=True, seed=0):
    """`num_samples` ancestral samples (num_samples x n* x p) of a dense GPAR conditioned on (x, y, w), at inputs xs - what
    `GPARRegressor.sample(xs, ws, posterior=True, num_samples=...)` draws (regression.py:508-564 over model.py:245-277): layer by layer a
    joint draw of the latent function at the current inputs from the conditioned layer, observation noise noise_i / w*_i on top, and
    the NOISY draw appended to the inputs of the next layer (with `replace` the posterior mean instead, model.py:291-322 with
    obs = None); the sample returned is the latent or the noisy draw.  Outputs un-normalised as regression.py:551-552.  Own random
    numbers (numpy's generator): comparable with the product's samples in distribution only."""
    rng = np.random.default_rng(seed)
    x = np.asarray(x, dtype=np.float64)
    x = x[:, None] if x.ndim == 1 else x
    xs0 = np.asarray(xs, dtype=np.float64)
    xs0 = xs0[:, None] if xs0.ndim == 1 else xs0
    y = np.asarray(y, dtype=np.float64)
    y = y[:, None] if y.ndim == 1 else y
    w = np.ones_like(y) if w is None else np.asarray(w, dtype=np.float64)
    m, p = x.shape[1], y.shape[1]
    ns = xs0.shape[0]
    ws = np.ones((ns, p)) if ws is None else np.asarray(ws, dtype=np.float64)
    offset, scale = _normalisation(y, normalise_y)
    y = (y - offset) / scale
    # the conditioning chain (model.py:116-149): per layer the training inputs, the solved weights and the inverse of the noisy Gram
    available = ~np.isnan(y)
    layers = []
    for i in range(p):
        mask = available[:, i].copy()
        if impute and i < p - 1:
            mask |= available[:, i + 1:].any(axis=1)
        x, yi, wi = x[mask], y[mask, i], w[mask, i]
        y, w, available = y[mask], w[mask], available[mask]
        spec, noise = layer_spec(hypers, m, i, config)
        have = ~np.isnan(yi)
        S = oracle_kernels.gram(spec, x[have], None, noise_diag=noise / wi[have], jitter=eps)
        Sinv = np.linalg.inv(S)
        layers.append((spec, noise, x[have].copy(), Sinv @ yi[have], Sinv))
        if i < p - 1:
            col = yi.copy()
            if impute or replace:
                mean_x = oracle_kernels.gram(spec, x, x[have]) @ layers[-1][3]
                if impute:
                    col[~have] = mean_x[~have]
                if replace:
                    col[have] = mean_x[have]
            x = np.concatenate([x, col[:, None]], axis=1)
    out = np.empty((num_samples, ns, p))
    for s_ in range(num_samples):
        xs = xs0
        for i, (spec, noise, xt, alpha, Sinv) in enumerate(layers):
            Ksx = oracle_kernels.gram(spec, xs, xt)
            mean = Ksx @ alpha
            cov = oracle_kernels.gram(spec, xs) - Ksx @ Sinv @ Ksx.T
            cov = 0.5 * (cov + cov.T) + 1e-10 * np.eye(ns)
            f = mean + np.linalg.cholesky(cov) @ rng.standard_normal(ns)
            noisy = f + np.sqrt(noise / ws[:, i]) * rng.standard_normal(ns)
            out[s_, :, i] = (f if latent else noisy) * scale[i] + offset[i]
            if i < p - 1:
                xs = np.concatenate([xs, (mean if replace else noisy)[:, None]], axis=1)
    return out


def bounds(name):
    """(lower, upper) of the optimiser's map for a hyper-parameter name, or None for an unconstrained one
    (regression.py:101-173: varz `bnd` defaults [1e-4, 1e4]; the RQ shapes [1e-3, 1e3]; the noise lower bound 1e-8; the additive
    constant of the input-linear term is `vs.get`, unconstrained)."""
    if name.endswith("/input/lin/const"):
        return None
    if name.endswith("/alpha"):
        return 1e-3, 1e3
    if name.endswith("/noise"):
        return 1e-8, 1e4
    return 1e-4, 1e4


def to_unconstrained(name, value, grad):
    """d / d(latent) from d / d(value) for varz's bounded map value = lower + (upper - lower) sigmoid(latent):
    d value / d latent = (value - lower)(upper - value) / (upper - lower)."""
    b = bounds(name)
    if b is None:
        return np.asarray(grad, dtype=np.float64)
    lo, hi = b
    value = np.asarray(value, dtype=np.float64)
    return np.asarray(grad, dtype=np.float64) * (value - lo) * (hi - value) / (hi - lo)


def fd_gradient(x, y, w, hypers, config, names=None, rel_step=1e-3, **kw):
    """{name: d gpar_logpdf / d hypers[name]} by fourth-order central differences, entry by entry:
    (-f(+2h) + 8 f(+h) - 8 f(-h) + f(-2h)) / 12h with h = rel_step * max(|value|, 1e-3).  The wide step is deliberate: with
    inducing points the value goes through K_zz^-1 at a jitter of 1e-12 and carries rounding noise of ~1e-8 relative, which a
    step of 1e-6 turns into 1e-4 of the gradient; at h = 1e-3 the noise term is 1e-7 and the truncation term (h^4) below it."""
    out = {}
    for name in (sorted(hypers) if names is None else names):
        base = np.array(hypers[name], dtype=np.float64)
        grad = np.zeros_like(base)
        flat = base.reshape(-1)
        for j in range(flat.size):
            h = rel_step * max(abs(flat[j]), 1e-3)
            vals = {}
            for k in (-2, -1, 1, 2):
                moved = flat.copy()
                moved[j] += k * h
                trial = dict(hypers)
                trial[name] = moved.reshape(base.shape)
                vals[k] = gpar_logpdf(x, y, w, trial, config, **kw)
            grad.reshape(-1)[j] = (-vals[2] + 8.0 * vals[1] - 8.0 * vals[-1] + vals[-2]) / (12.0 * h)
        out[name] = grad
    return out
