"""Oracle (TEST INFRASTRUCTURE): the reference's CPU path restated on the very torch-CPU fp64 operators `lab.torch`
dispatches to - the `cpu_baseline` of bench.py (kind "port") and the second distance formula of the parity tests.

The reference evaluates a layer as stheno -> mlkernels -> lab -> torch (gpar/regression.py:5,8 import `lab.torch`):
  * one n x n temporary per kernel term and factor, never fused (mlkernels `Sum` / `Product` / `Scaled` nodes;
    gpar/regression.py:110,127-129,138,146,166);
  * pairwise squared distances by lab's `pw_dists2`: for one feature `(a - b^T)^2`, otherwise the expansion
    `|a|^2 + |b|^2 - 2 a b^T` - NOT the explicit differences the HIP kernel and
    oracle/kernels.py use; `tests/test_oracle.py` bounds what that difference does to a log marginal likelihood;
  * `torch.linalg.cholesky` of K + diag(noise) + 1e-12 I (lab's `B.epsilon`), `torch.linalg.solve_triangular`;
  * `fit`: torch autograd through all of the above (varz.torch.minimise_l_bfgs_b, gpar/regression.py:459);
  * `predict`: per sample and per layer a cross-Gram, a triangular solve against the training factor, an n* x n*
    Cholesky and a matrix-vector product (gpar/regression.py:559-563, gpar/model.py:259-275).
Every function returns its result together with a dict of wall-clock seconds per stage.
"""
import time

import numpy as np
import torch

__all__ = ["set_threads", "cpu_quota", "gram", "layer_logpdf", "layer_vfe_bound", "layer_objective_and_gradient", "layer_posterior_sample",
           "leaf_spec", "layer_fit"]

EPSILON = 1e-12  # lab's B.epsilon


def cpu_quota():
    """CPUs the container may use per scheduling period (cgroup v2 cpu.max / v1 cfs quota), or None without a limit."""
    import math

    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else max(1, math.ceil(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    try:
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if quota <= 0 else max(1, math.ceil(quota / period))
    except (OSError, ValueError):
        return None


def set_threads(threads=None):
    """All the CPUs this process may actually use: the host's logical CPUs, capped by the container's CPU quota (more
    threads than that only get the process throttled: the worker threads of every parallel region spin)."""
    import os

    if not threads:
        threads = os.cpu_count() or 1
        quota = cpu_quota()
        if quota:
            threads = min(threads, quota)
    torch.set_num_threads(int(threads))
    return torch.get_num_threads()


def _features(factor, x):
    cols = list(factor["cols"])
    sel = x[:, cols] if cols else x.new_zeros((x.shape[0], 0))
    if factor.get("periods") is not None:
        freq = 2.0 * np.pi / torch.as_tensor(factor["periods"], dtype=torch.float64)
        sel = torch.cat([torch.sin(sel * freq[None, :]), torch.cos(sel * freq[None, :])], dim=1)
    scales = factor["scales"]
    scales = scales if isinstance(scales, torch.Tensor) else torch.as_tensor(scales, dtype=torch.float64)
    return sel / scales[None, :]


def _pw_dists2(a, b):
    """lab's pairwise squared distances: explicit difference for a single feature, the norm expansion otherwise."""
    if a.shape[1] == 1:
        return (a - b.T) ** 2
    na = torch.sum(a * a, dim=1)[:, None]
    nb = torch.sum(b * b, dim=1)[None, :]
    return na + nb - 2.0 * (a @ b.T)


def _factor_matrix(factor, x1, x2):
    z1, z2 = _features(factor, x1), _features(factor, x2)
    if factor["type"] == "linear":
        return z1 @ z2.T
    r2 = _pw_dists2(z1, z2)
    if factor["type"] == "eq":
        return torch.exp(-0.5 * r2)
    alpha = factor["alpha"]
    return (1.0 + r2 / (2.0 * alpha)) ** (-alpha)


def gram(spec, x1, x2=None):
    """K(x1, x2) term by term, one temporary per node of the kernel expression (unfused, as mlkernels evaluates it)."""
    x2 = x1 if x2 is None else x2
    out = None
    for term in spec["terms"]:
        prod = None
        for factor in term["factors"]:
            mat = _factor_matrix(factor, x1, x2)
            prod = mat if prod is None else prod * mat
        coef = term["coef"]
        if prod is None:
            prod = torch.ones(x1.shape[0], x2.shape[0], dtype=torch.float64)
        prod = coef * prod
        out = prod if out is None else out + prod
    if out is None:
        out = torch.zeros(x1.shape[0], x2.shape[0], dtype=torch.float64)
    return out


def _as_torch(a):
    return a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, dtype=np.float64))


def layer_logpdf(spec, x, y, noise_diag):
    """log N(y; 0, K + diag(noise)) of one layer; returns (value, seconds per stage, factor L)."""
    x, y, noise_diag = _as_torch(x), _as_torch(y).reshape(-1, 1), _as_torch(noise_diag).reshape(-1)
    n = x.shape[0]
    t0 = time.perf_counter()
    K = gram(spec, x)
    K = K + torch.diag(noise_diag + EPSILON)
    t1 = time.perf_counter()
    L = torch.linalg.cholesky(K)
    t2 = time.perf_counter()
    z = torch.linalg.solve_triangular(L, y, upper=False)
    value = -0.5 * (2.0 * torch.sum(torch.log(torch.diagonal(L))) + n * np.log(2.0 * np.pi) + torch.sum(z * z))
    t3 = time.perf_counter()
    return float(value), {"gram_s": t1 - t0, "potrf_s": t2 - t1, "solve_s": t3 - t2}, L


def layer_objective_and_gradient(make_spec, params, x, y, noise_index=None):
    """One evaluation of `fit`'s objective and its autograd gradient (the reference differentiates through the Gram
    build, the Cholesky and the solve).  `make_spec(params)` must build the kernel dict from the list of leaf tensors
    `params` (all requiring grad); `params[noise_index]` is the noise variance."""
    x, y = _as_torch(x), _as_torch(y).reshape(-1, 1)
    n = x.shape[0]
    leaves = [p.detach().clone().requires_grad_(True) for p in params]
    t0 = time.perf_counter()
    spec = make_spec(leaves)
    K = gram(spec, x) + torch.diag((leaves[noise_index] + EPSILON).expand(n))
    L = torch.linalg.cholesky(K)
    z = torch.linalg.solve_triangular(L, y, upper=False)
    value = 0.5 * (2.0 * torch.sum(torch.log(torch.diagonal(L))) + n * np.log(2.0 * np.pi) + torch.sum(z * z))
    t1 = time.perf_counter()
    value.backward()
    t2 = time.perf_counter()
    return float(value.detach()), [p.grad for p in leaves], {"forward_s": t1 - t0, "backward_s": t2 - t1}


def layer_posterior_sample(spec, x, L, z, x_star, noise_star, generator=None):
    """One posterior draw of one layer at x_star given the training factor L and z = L^-1 y:
    mean = V^T z, cov = K** - V^T V with V = L^-1 K(X, x*), sample = mean + chol(cov + noise + eps) randn."""
    x, x_star = _as_torch(x), _as_torch(x_star)
    t0 = time.perf_counter()
    Kxs = gram(spec, x, x_star)
    Kss = gram(spec, x_star)
    t1 = time.perf_counter()
    V = torch.linalg.solve_triangular(L, Kxs, upper=False)
    mean = V.T @ z
    cov = Kss - V.T @ V + torch.diag(_as_torch(noise_star).reshape(-1) + EPSILON)
    t2 = time.perf_counter()
    Ls = torch.linalg.cholesky(cov)
    draw = mean + Ls @ torch.randn(x_star.shape[0], 1, dtype=torch.float64, generator=generator)
    t3 = time.perf_counter()
    return draw, {"gram_s": t1 - t0, "solve_s": t2 - t1, "potrf_s": t3 - t2}


def layer_vfe_bound(spec, x, y, noise_diag, z):
    """Titsias' bound of one layer with inducing inputs z, in the order stheno's PseudoObs evaluates it:
    L_z = chol(K_zz + eps I), B = L_z^-1 K_zx D^-1/2, A = I + B B^T, c = B D^-1/2 y;  returns (value, seconds per stage)."""
    x, z = _as_torch(x), _as_torch(z)
    y, d = _as_torch(y).reshape(-1, 1), _as_torch(noise_diag).reshape(-1)
    n, M = x.shape[0], z.shape[0]
    t0 = time.perf_counter()
    Kzz = gram(spec, z) + EPSILON * torch.eye(M, dtype=torch.float64)
    Kzx = gram(spec, z, x)
    # diagonal of K_xx: the kernels here are stationary + linear; evaluated row by row it would be n tiny launches - mlkernels has an
    # elementwise path for it, restated: sum over terms of coef * prod(factor(x_i, x_i))
    diag = torch.zeros(n, dtype=torch.float64)
    for term in spec["terms"]:
        prod = torch.ones(n, dtype=torch.float64)
        for factor in term["factors"]:
            if factor["type"] == "linear":
                zf = _features(factor, x)
                prod = prod * torch.sum(zf * zf, dim=1)
        diag = diag + term["coef"] * prod
    t1 = time.perf_counter()
    Lz = torch.linalg.cholesky(Kzz)
    B = torch.linalg.solve_triangular(Lz, Kzx, upper=False) / torch.sqrt(d)[None, :]
    t2 = time.perf_counter()
    A = torch.eye(M, dtype=torch.float64) + B @ B.T
    c = B @ (y / torch.sqrt(d)[:, None])
    t3 = time.perf_counter()
    La = torch.linalg.cholesky(A)
    v = torch.linalg.solve_triangular(La, c, upper=False)
    trace = torch.sum(diag / d) - torch.sum(B * B)
    value = -0.5 * (2.0 * torch.sum(torch.log(torch.diagonal(La))) + torch.sum(torch.log(2.0 * np.pi * d)) + torch.sum(y * y / d[:, None])
                    - torch.sum(v * v) + trace)
    t4 = time.perf_counter()
    layer_vfe_bound.last_factors = (Lz, La, v)   # (for layer_vfe_posterior_sample: what conditioning on the same data would hold)
    return float(value), {"gram_s": t1 - t0, "solve_s": t2 - t1, "product_s": t3 - t2, "potrf_s": t4 - t3}


def layer_vfe_posterior_sample(spec, z, Lz, La, v, x_star, noise_star, generator=None):
    """One posterior draw of one inducing-point layer at x_star (stheno's PseudoObs posterior; reference
    gpar/model.py:286-287 then :264-270): with P = K(x*, Z) L_z^-T and Q = P L_A^-T,
    mean = Q (L_A^-1 c) (`v`), cov = K** - P P^T + Q Q^T, sample = mean + chol(cov + noise + eps) randn.  Returns (draw, seconds per stage)."""
    z, x_star = _as_torch(z), _as_torch(x_star)
    t0 = time.perf_counter()
    Ksz = gram(spec, x_star, z)
    Kss = gram(spec, x_star)
    t1 = time.perf_counter()
    P = torch.linalg.solve_triangular(Lz, Ksz.T, upper=False).T
    Q = torch.linalg.solve_triangular(La, P.T, upper=False).T
    mean = Q @ v
    cov = Kss - P @ P.T + Q @ Q.T + torch.diag(_as_torch(noise_star).reshape(-1) + EPSILON)
    t2 = time.perf_counter()
    Ls = torch.linalg.cholesky(cov)
    draw = mean + Ls @ torch.randn(x_star.shape[0], 1, dtype=torch.float64, generator=generator)
    t3 = time.perf_counter()
    return draw, {"gram_s": t1 - t0, "solve_s": t2 - t1, "potrf_s": t3 - t2}


def leaf_spec(spec):
    """(leaves, rebuild): the kernel dict's coefficients and length scales as torch leaves for autograd."""
    leaves = []
    for term in spec["terms"]:
        leaves.append(torch.tensor(float(term["coef"]), dtype=torch.float64))
        for factor in term["factors"]:
            leaves.append(torch.tensor(list(factor["scales"]), dtype=torch.float64))

    def rebuild(params):
        it = iter(params)
        out = {"terms": []}
        for term in spec["terms"]:
            coef = next(it)
            factors = [dict(factor, scales=next(it)) for factor in term["factors"]]
            out["terms"].append({"coef": coef, "factors": factors})
        return out

    return leaves, rebuild


def layer_fit(spec, noise, x, y, iters=20):
    """`fit` of ONE layer as the reference runs it (regression.py:434-459): scipy's L-BFGS-B (what varz.minimise_l_bfgs_b
    drives) over the logarithms of the layer's coefficients, length scales and noise, objective and gradient by torch autograd
    through Gram, Cholesky and solve.  Returns (final objective, evaluations, seconds)."""
    import scipy.optimize

    leaves, rebuild = leaf_spec(spec)
    leaves.append(torch.tensor(float(noise), dtype=torch.float64))
    shapes = [tuple(leaf.shape) for leaf in leaves]
    x0 = np.concatenate([np.log(np.abs(leaf.numpy().reshape(-1)) + 1e-300) for leaf in leaves])
    signs = np.concatenate([np.sign(leaf.numpy().reshape(-1)) + (leaf.numpy().reshape(-1) == 0) for leaf in leaves])
    count = [0]

    def fg(u):
        count[0] += 1
        vals = signs * np.exp(u)
        params, at = [], 0
        for shape in shapes:
            size = int(np.prod(shape)) if shape else 1
            params.append(torch.tensor(vals[at:at + size].reshape(shape), dtype=torch.float64))
            at += size
        try:
            value, grads, _ = layer_objective_and_gradient(lambda ps: rebuild(ps[:-1]), params, x, y, noise_index=-1)
        except Exception:  # noqa: BLE001 - a trial point of the line search where the Cholesky fails: varz reports NaN there; a
            return 1e30, np.zeros_like(u)   # large finite value makes scipy's line search back off the same way
        g = np.concatenate([gr.numpy().reshape(-1) for gr in grads]) * vals
        if not np.isfinite(value) or not np.all(np.isfinite(g)):
            return 1e30, np.zeros_like(u)
        return value, g

    t0 = time.perf_counter()
    _, final, _ = scipy.optimize.fmin_l_bfgs_b(fg, x0, maxiter=iters)
    return float(final), count[0], time.perf_counter() - t0
