"""Oracle (TEST INFRASTRUCTURE): Gram matrices of GPAR's composite layer kernels in numpy fp64.

Restates the public definitions of the mlkernels kernels that the reference composes in
/root/reference/gpar/regression.py:92-180 (the packages themselves are not vendored; see
oracle/__init__.py, "parity unpinned"):

    EQ            k(x, y) = exp(-|x - y|^2 / 2)                     (regression.py:109,128,165)
    RQ(alpha)     k(x, y) = (1 + |x - y|^2 / (2 alpha))^-alpha      (regression.py:107,156-163)
    Linear        k(x, y) = <x, y>                                  (regression.py:138,146)
    k.stretch(s)  k(x / s, y / s)                                   (regression.py:110,128,138,146,166)
    k.periodic(T) k(phi(x), phi(y)), phi(x) = [sin(2 pi x / T), cos(2 pi x / T)]  (regression.py:128)
    k.select(i)   k(x[:, i], y[:, i])                               (regression.py:178)
    c * k, k + k', k * k', k + c                                    (regression.py:110,127-129,138)

A kernel is handed around as a plain dict (no product classes are imported here):

    {"terms": [{"coef": float,
                "factors": [{"type": "eq" | "rq" | "linear",
                             "cols": [int, ...],        # columns of the design matrix
                             "scales": [float, ...],    # one per feature (2 * len(cols) if periodic)
                             "periods": None | [float, ...],   # one per column
                             "alpha": float}]}]}

The squared distance is computed from explicit differences, sum_d ((x_d - y_d) / s_d)^2, which is exactly
symmetric and exactly zero on the diagonal; the HIP kernel uses the same formula.
"""
import numpy as np

__all__ = ["features", "factor_matrix", "gram", "gram_diag", "spec_to_dict"]


def features(factor, x):
    """Feature rows of one factor: select -> (periodic embedding) -> stretch."""
    x = np.asarray(x, dtype=np.float64)
    cols = list(factor["cols"])
    sel = x[:, cols] if len(cols) else np.zeros((x.shape[0], 0))
    periods = factor.get("periods")
    if periods is not None:
        freq = 2.0 * np.pi / np.asarray(periods, dtype=np.float64)
        sel = np.concatenate([np.sin(sel * freq[None, :]), np.cos(sel * freq[None, :])], axis=1)
    scales = np.asarray(factor["scales"], dtype=np.float64)
    if sel.shape[1] != scales.shape[0]:
        raise ValueError("scales do not match the number of features")
    return sel * (1.0 / scales)[None, :]


def factor_matrix(factor, x1, x2):
    z1, z2 = features(factor, x1), features(factor, x2)
    if factor["type"] == "linear":
        out = np.zeros((z1.shape[0], z2.shape[0]))
        for d in range(z1.shape[1]):
            out += z1[:, d][:, None] * z2[:, d][None, :]
        return out
    r2 = np.zeros((z1.shape[0], z2.shape[0]))
    for d in range(z1.shape[1]):
        diff = z1[:, d][:, None] - z2[:, d][None, :]
        r2 += diff * diff
    if factor["type"] == "eq":
        return np.exp(-0.5 * r2)
    if factor["type"] == "rq":
        alpha = float(factor["alpha"])
        return np.exp(-alpha * np.log1p(r2 / (2.0 * alpha)))
    raise ValueError(f"unknown factor type {factor['type']!r}")


def gram(spec, x1, x2=None, noise_diag=None, jitter=0.0):
    """K[a, b] = k(x1[a], x2[b]); for x2 None the symmetric Gram, optionally + diag(noise_diag) + jitter I."""
    x1 = np.asarray(x1, dtype=np.float64)
    sym = x2 is None
    x2 = x1 if sym else np.asarray(x2, dtype=np.float64)
    out = np.zeros((x1.shape[0], x2.shape[0]))
    for term in spec["terms"]:
        prod = np.full_like(out, float(term["coef"]))
        for factor in term["factors"]:
            prod = prod * factor_matrix(factor, x1, x2)
        out += prod
    if sym:
        idx = np.arange(out.shape[0])
        if noise_diag is not None:
            out[idx, idx] += np.asarray(noise_diag, dtype=np.float64)
        if jitter:
            out[idx, idx] += jitter
    return out


def gram_diag(spec, x):
    x = np.asarray(x, dtype=np.float64)
    out = np.zeros(x.shape[0])
    for term in spec["terms"]:
        prod = np.full_like(out, float(term["coef"]))
        for factor in term["factors"]:
            if factor["type"] == "linear":
                z = features(factor, x)
                prod = prod * np.sum(z * z, axis=1)
        out += prod
    return out


def spec_to_dict(kernel):
    """Duck-typed conversion of a product-side kernel object (anything exposing `.terms` with
    `.coef`/`.factors`, factors exposing type/cols/scales/periods/alpha) into the oracle's dict."""
    if isinstance(kernel, dict):
        return kernel
    terms = []
    for term in kernel.terms:
        factors = []
        for f in term.factors:
            factors.append(
                {
                    "type": str(f.type),
                    "cols": [int(c) for c in f.cols],
                    "scales": [float(s) for s in np.asarray(f.scales_value(), dtype=np.float64).reshape(-1)],
                    "periods": None
                    if f.periods is None
                    else [float(s) for s in np.asarray(f.periods_value(), dtype=np.float64).reshape(-1)],
                    "alpha": float(f.alpha_value()) if f.alpha is not None else 0.0,
                }
            )
        terms.append({"coef": float(term.coef_value()), "factors": factors})
    return {"terms": terms}
