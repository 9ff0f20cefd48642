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

__all__ = ["set_threads", "cpu_quota", "gram", "layer_logpdf", "layer_objective_and_gradient", "layer_posterior_sample"]

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
