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


def kernel_grads(spec, x, W):
    """d/d(theta) of  1/2 sum_ab W_ab K_ab(theta)  for every parameter of the kernel `spec` (W symmetric, n x n):
    term coefficients, per-feature length scales, per-column periods and RQ alphas.  Each derivative matrix
    dK/dtheta is formed explicitly (test sizes only) — deliberately a different route from the HIP kernel, which
    accumulates per-feature moment sums in one pass.  Verified against central finite differences in
    tests/test_oracle.py."""
    x = np.asarray(x, dtype=np.float64)
    W = np.asarray(W, dtype=np.float64)
    out = {"coef": [], "factors": []}
    for term in spec["terms"]:
        mats = [factor_matrix(f, x, x) for f in term["factors"]]
        full = np.ones((x.shape[0], x.shape[0]))
        for m_ in mats:
            full = full * m_
        out["coef"].append(0.5 * np.sum(W * full))
        fgrads = []
        for fi, f in enumerate(term["factors"]):
            rest = np.full_like(full, float(term["coef"]))
            for fj, m_ in enumerate(mats):
                if fj != fi:
                    rest = rest * m_
            z = features(f, x)
            scales = np.asarray(f["scales"], dtype=np.float64)
            nd = z.shape[1]
            g = {"scales": np.zeros(nd), "periods": None, "alpha": None}
            if f["type"] == "linear":
                for q in range(nd):
                    dF = -2.0 * (z[:, q][:, None] * z[:, q][None, :]) / scales[q]
                    g["scales"][q] = 0.5 * np.sum(W * rest * dF)
            else:
                r2 = np.zeros_like(full)
                diffs = []
                for q in range(nd):
                    d = z[:, q][:, None] - z[:, q][None, :]
                    diffs.append(d)
                    r2 += d * d
                if f["type"] == "eq":
                    F = np.exp(-0.5 * r2)
                    dF_dr2 = -0.5 * F
                else:
                    alpha = float(f["alpha"])
                    F = np.exp(-alpha * np.log1p(r2 / (2 * alpha)))
                    dF_dr2 = -0.5 * F / (1.0 + r2 / (2 * alpha))
                    t = r2 / (2 * alpha)
                    g["alpha"] = 0.5 * np.sum(W * rest * F * (t / (1 + t) - np.log1p(t)))
                for q in range(nd):
                    # d r2 / d s_q = -2 diff_q^2 / s_q
                    g["scales"][q] = 0.5 * np.sum(W * rest * dF_dr2 * (-2.0 * diffs[q] ** 2 / scales[q]))
                if f.get("periods") is not None:
                    periods = np.asarray(f["periods"], dtype=np.float64)
                    cols = list(f["cols"])
                    ncol = len(cols)
                    g["periods"] = np.zeros(ncol)
                    for j in range(ncol):
                        omega = 2 * np.pi / periods[j]
                        xc = x[:, cols[j]]
                        # z_sin = sin(omega x)/s, z_cos = cos(omega x)/s';  d omega / d T = -omega / T
                        dz_sin = np.cos(omega * xc) * xc * (-omega / periods[j]) / scales[j]
                        dz_cos = -np.sin(omega * xc) * xc * (-omega / periods[j]) / scales[j + ncol]
                        dr2 = 2 * diffs[j] * (dz_sin[:, None] - dz_sin[None, :]) + 2 * diffs[j + ncol] * (
                            dz_cos[:, None] - dz_cos[None, :]
                        )
                        g["periods"][j] = 0.5 * np.sum(W * rest * dF_dr2 * dr2)
            fgrads.append(g)
        out["factors"].append(fgrads)
    return out


def kernel_input_grads(spec, x1, x2, W):
    """G[a, c] = sum_b W[a, b] d k(x1_a, x2_b) / d x1[a, c]  (x2 held fixed), through explicit per-column derivative matrices
    of every factor (test sizes only) - a different route from the HIP kernel, which works in feature space with row sums."""
    x1, x2, W = np.asarray(x1, dtype=np.float64), np.asarray(x2, dtype=np.float64), np.asarray(W, dtype=np.float64)
    n1, width = x1.shape
    out = np.zeros((n1, width))
    for term in spec["terms"]:
        mats = [factor_matrix(f, x1, x2) for f in term["factors"]]
        for fi, f in enumerate(term["factors"]):
            rest = np.full((n1, x2.shape[0]), float(term["coef"]))
            for fj, m_ in enumerate(mats):
                if fj != fi:
                    rest = rest * m_
            z1, z2 = features(f, x1), features(f, x2)
            cols = list(f["cols"])
            scales = np.asarray(f["scales"], dtype=np.float64)
            periods = f.get("periods")
            ncol = len(cols)
            if f["type"] != "linear":
                r2 = np.zeros_like(rest)
                for q in range(z1.shape[1]):
                    r2 += (z1[:, q][:, None] - z2[:, q][None, :]) ** 2
                if f["type"] == "eq":
                    dF_dr2 = -0.5 * np.exp(-0.5 * r2)
                else:
                    alpha = float(f["alpha"])
                    dF_dr2 = -0.5 * np.exp(-alpha * np.log1p(r2 / (2 * alpha))) / (1.0 + r2 / (2 * alpha))
            for q in range(z1.shape[1]):
                j = q % ncol if ncol else 0
                c = cols[j]
                # d z1[:, q] / d x1[:, c]
                if periods is None:
                    dz = np.full(n1, 1.0 / scales[q])
                else:
                    omega = 2 * np.pi / float(periods[j])
                    dz = (omega * np.cos(omega * x1[:, c]) if q < ncol else -omega * np.sin(omega * x1[:, c])) / scales[q]
                if f["type"] == "linear":
                    dF = z2[:, q][None, :] * np.ones((n1, 1))            # d <z1, z2> / d z1[q]
                else:
                    dF = dF_dr2 * 2.0 * (z1[:, q][:, None] - z2[:, q][None, :])
                out[:, c] += np.sum(W * rest * dF, axis=1) * dz
    return out
