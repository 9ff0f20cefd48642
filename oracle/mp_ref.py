"""Oracle (TEST INFRASTRUCTURE): the same closed forms as oracle/gp_ref.py and oracle/gpar_ref.py in 50-digit arithmetic
(mpmath), to bound the floating-point error of the fp64 oracle routes that produced the golden vectors.  Tiny problems
only (n <= 30: a few seconds).  Not a second opinion on the FORMULAS - those are pinned by closed forms, finite
differences and scikit-learn (oracle/__init__.py) - but on their evaluation: whatever the fp64 routes lose to
cancellation or conditioning shows up as a difference from these values.
"""
import mpmath as mp

mp.mp.dps = 50

__all__ = ["gram", "logpdf", "posterior", "vfe_bound", "gpar_logpdf"]


def _features(factor, x):
    cols = list(factor["cols"])
    rows = []
    periods = factor.get("periods")
    scales = [mp.mpf(s) for s in factor["scales"]]
    for row in x:
        sel = [mp.mpf(row[c]) for c in cols]
        if periods is not None:
            freq = [2 * mp.pi / mp.mpf(t) for t in periods]
            sel = [mp.sin(v * f) for v, f in zip(sel, freq)] + [mp.cos(v * f) for v, f in zip(sel, freq)]
        rows.append([v / s for v, s in zip(sel, scales)])
    return rows


def _factor_entry(factor, za, zb):
    if factor["type"] == "linear":
        return mp.fsum(a * b for a, b in zip(za, zb))
    r2 = mp.fsum((a - b) ** 2 for a, b in zip(za, zb))
    if factor["type"] == "eq":
        return mp.exp(-r2 / 2)
    alpha = mp.mpf(factor["alpha"])
    return (1 + r2 / (2 * alpha)) ** (-alpha)


def gram(spec, x1, x2=None, noise_diag=None, jitter=0):
    sym = x2 is None
    x2 = x1 if sym else x2
    n1, n2 = len(x1), len(x2)
    out = mp.zeros(n1, n2)
    for term in spec["terms"]:
        feats = [(f, _features(f, x1), _features(f, x2)) for f in term["factors"]]
        coef = mp.mpf(term["coef"])
        for a in range(n1):
            for b in range(n2):
                v = coef
                for f, z1, z2 in feats:
                    v *= _factor_entry(f, z1[a], z2[b])
                out[a, b] += v
    if sym:
        for a in range(n1):
            if noise_diag is not None:
                out[a, a] += mp.mpf(noise_diag[a])
            out[a, a] += mp.mpf(jitter)
    return out


def _mvn_logpdf(S, r):
    L = mp.cholesky(S)
    z = mp.lu_solve(L, r)  # L is triangular: exact elimination
    logdet = 2 * mp.fsum(mp.log(L[i, i]) for i in range(S.rows))
    return -(logdet + S.rows * mp.log(2 * mp.pi) + mp.fsum(v * v for v in z)) / 2


def _solve(S, B):
    """S^-1 B for a matrix B, column by column."""
    out = mp.zeros(B.rows, B.cols)
    for j in range(B.cols):
        col = mp.lu_solve(S, B[:, j])
        for i in range(B.rows):
            out[i, j] = col[i]
    return out


def _col(v):
    return mp.matrix([mp.mpf(float(a)) for a in v])


def logpdf(spec, x, y, noise, eps=1e-12):
    n = len(y)
    noise = [float(noise)] * n if not hasattr(noise, "__len__") else list(noise)
    return _mvn_logpdf(gram(spec, x, None, noise, eps), _col(y))


def posterior(spec, x, y, noise, xs, eps=1e-12):
    n = len(y)
    noise = [float(noise)] * n if not hasattr(noise, "__len__") else list(noise)
    S = gram(spec, x, None, noise, eps)
    Ksx = gram(spec, xs, x)
    mean = Ksx * mp.lu_solve(S, _col(y))
    cov = gram(spec, xs) - Ksx * _solve(S, Ksx.T)
    return mean, cov


def vfe_bound(spec, x, y, noise, z, eps=1e-12):
    n = len(y)
    d = [mp.mpf(float(noise))] * n if not hasattr(noise, "__len__") else [mp.mpf(float(v)) for v in noise]
    Kzz = gram(spec, z, None, None, eps)
    Kxz = gram(spec, x, z)
    Q = Kxz * _solve(Kzz, Kxz.T)
    S = Q.copy()
    for a in range(n):
        S[a, a] += d[a]
    kdiag = [gram(spec, [x[a]])[0, 0] for a in range(n)]
    return _mvn_logpdf(S, _col(y)) - mp.fsum((kdiag[a] - Q[a, a]) / d[a] for a in range(n)) / 2


def gpar_logpdf(x, y, w, hypers, config, impute=False, replace=False, eps=1e-12):
    """oracle/gpar_ref.gpar_logpdf in 50 digits (dense path; NaN = missing)."""
    import numpy as np

    from . import gpar_ref

    x = np.asarray(x, dtype=np.float64)
    x = x[:, None] if x.ndim == 1 else x
    y = np.asarray(y, dtype=np.float64)
    w = np.ones_like(y) if w is None else np.asarray(w, dtype=np.float64)
    m, p = x.shape[1], y.shape[1]
    rows = [[mp.mpf(float(v)) for v in row] for row in x]
    available = ~np.isnan(y)
    total = mp.mpf(0)
    for i in range(p):
        mask = available[:, i].copy()
        if impute and i < p - 1:
            mask |= available[:, i + 1:].any(axis=1)
        rows = [r for r, keep in zip(rows, mask) if keep]
        yi, wi = y[mask, i], w[mask, i]
        y, w, available = y[mask], w[mask], available[mask]
        spec, noise = gpar_ref.layer_spec(hypers, m, i, config)
        have = ~np.isnan(yi)
        xh = [r for r, h in zip(rows, have) if h]
        nh = [mp.mpf(noise) / mp.mpf(float(v)) for v in wi[have]]
        total += _mvn_logpdf(gram(spec, xh, None, nh, eps), _col(yi[have]))
        if i < p - 1:
            col = [mp.mpf(float(v)) if h else None for v, h in zip(yi, have)]
            if (impute and (~have).any()) or (replace and have.any()):
                S = gram(spec, xh, None, nh, eps)
                mean = gram(spec, rows, xh) * mp.lu_solve(S, _col(yi[have]))
                for k, h in enumerate(have):
                    if (impute and not h) or (replace and h):
                        col[k] = mean[k]
            rows = [r + [c if c is not None else mp.nan] for r, c in zip(rows, col)]
    return total
