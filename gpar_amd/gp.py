"""GP layer objects: the stheno subset GPAR drives, computed on the MI355X through the engine seam.

The reference treats a layer as `(GP, noise)` and uses exactly these operations on it
(/root/reference/gpar/model.py): `f(x, noise)` -> finite-dimensional distribution (:264,270,287,289),
`Obs(fdd, y)` / `PseudoObs(f(x_ind), fdd, y)` (:286-289), `f.measure.logpdf(obs)` (:226), `f | obs` (:170,232,298),
`.mean(x)` (:299,301), `fdd.sample()` (:235,264,270) and, in its tests, `fdd.logpdf(y)` and `f | (fdd, y)`.
This module provides those names with the same call signatures; the arithmetic is

  exact GP (Rasmussen & Williams alg. 2.1), evaluated with ONE partial Cholesky of an augmented matrix

        [ K + D + eps I      .  ]   factor the first n columns     [ L          . ]
        [ y^T                0  ]   ------------------------>      [ (L^-1 y)^T   -|L^-1 y|^2 ]

    which yields log|K|, the quadratic form and L in a single device pipeline (include/gpar_hip.h:
    gpar_potrf); posterior moments at new inputs are V = K_*x L^-T (TRSM), mean = V (L^-1 y),
    cov = K_** - V V^T (SYRK);

  inducing points (Titsias 2009, VFE; stheno's PseudoObs default):  L_z = chol(K_zz), B^T = K_xz L_z^-T,
    A = I + B D^-1 B^T, bound = -1/2 [ sum_j (k_jj - |B_:j|^2)/d_j + sum_j log(2 pi d_j) + log|A| + y^T D^-1 y - |L_A^-1 B D^-1 y|^2 ].

All heavy steps are engine primitives (HIP kernels); torch is used here only for O(n) vector plumbing.
"""
import contextlib
import math

import numpy as np
import torch

from .engine import get_engine, joining
from .kernels import Kernel

__all__ = ["Measure", "GP", "FDD", "Obs", "PseudoObs", "SparseObs"]

_LOG_2PI = math.log(2.0 * math.pi)


def _as_matrix(eng, x):
    """Engine tensor of rank 2 (`B.uprank`: a vector becomes a column)."""
    t = eng.tensor(x)
    if t.dim() == 0:
        t = t.reshape(1, 1)
    elif t.dim() == 1:
        t = t[:, None]
    return t


def _noise_vector(eng, noise, n):
    """None | scalar | length-n vector -> device vector of length n (or None)."""
    if noise is None:
        return None
    t = eng.tensor(noise)
    if t.dim() == 0 or t.numel() == 1:
        return t.reshape(1).expand(n).contiguous()
    t = t.reshape(-1)
    if t.numel() != n:
        raise ValueError(f"noise has {t.numel()} entries for {n} inputs")
    return t.contiguous()


class _Factor:
    """Cholesky of an n x n SPD matrix together with the solve against one right-hand side, from one partial
    factorisation of the (n + 1) x (n + 1) augmented matrix.  `fill(block)` must write the lower triangle of the
    SPD matrix into `block` (an n x n view)."""

    def __init__(self, eng, n, fill, rhs):
        self.eng, self.n = eng, n
        A = eng.new_matrix(n + 1, n + 1)
        if n > 0:
            fill(A[:n, :n])
            A[n, :n] = rhs.reshape(-1)
        A[n, n] = 0.0
        if n > 0:
            logdet, info = eng.potrf_(A, nf=n)
            eng.check_info(info)
        else:
            logdet = torch.zeros(1, dtype=torch.float64, device=A.device)
        self.A = A
        self.L = A[:n, :n]
        self.zrow = A[n : n + 1, :n]  # (L^-1 rhs)^T, 1 x n
        self.logdet = logdet
        self.quad = -A[n, n]  # |L^-1 rhs|^2 (device scalar)
        self._alpha = None

    @classmethod
    def placeholder(cls, eng, n):
        """Buffers of a factor that another process computes (parallel.sharded_condition receives into `A`)."""
        self = cls.__new__(cls)
        self.eng, self.n = eng, n
        self.A = eng.new_matrix(n + 1, n + 1)
        self.L = self.A[:n, :n]
        self.zrow = self.A[n : n + 1, :n]
        self.logdet = None  # not exchanged: a received factor serves posterior means / samples, not logpdf
        self.quad = None
        self._alpha = None
        return self

    def logpdf(self):
        """-1/2 (log|S| + n log 2 pi + rhs^T S^-1 rhs) as a 0-d tensor: on the CPU (synchronises), or left on the
        device while the engine is deferring checks (the caller then reads the sum of many layers once)."""
        val = (-0.5 * (self.logdet[0] + self.n * _LOG_2PI + self.quad)).detach()
        return val if getattr(self.eng, "_deferred", None) is not None else val.cpu()

    def alpha(self):
        """S^-1 rhs as a row (1 x n): alpha^T = z^T L^-1."""
        if self._alpha is None:
            a = self.zrow.clone()
            if self.n > 0:
                self.eng.trsm_rln_(self.L, a)
            self._alpha = a
        return self._alpha


def _needs_grad(v):
    return isinstance(v, torch.Tensor) and v.requires_grad


def kernel_parameters(kernel):
    """(kind, term index, factor index, tensor) for every kernel parameter that is a torch tensor requiring grad."""
    out = []
    for ti, term in enumerate(kernel.terms):
        if _needs_grad(term.coef):
            out.append(("coef", ti, None, term.coef))
        for fi, f in enumerate(term.factors):
            for kind in ("scales", "periods", "alpha"):
                v = getattr(f, kind)
                if _needs_grad(v):
                    out.append((kind, ti, fi, v))
    return out


class _LogMarginal(torch.autograd.Function):
    """log N(y; 0, K_theta + D) - or, for inducing-point observations, the VFE bound - as a differentiable function of
    the kernel parameters and the noise vector.

    Forward is the fused Gram + augmented Cholesky on the device.  Backward is analytic:
    with W = alpha alpha^T - K^-1,  d/dtheta = 1/2 sum_ab W_ab dK_ab/dtheta; K^-1 comes from L (TRSM on the
    identity + SYRK) and all kernel-parameter sums are produced by one fused pass over W on the device.
    torch then chains these through whatever produced the parameters (bound transforms, products, noise / w).
    """

    @staticmethod
    def forward(ctx, obs, noise, *tensors):
        ctx.obs = obs
        ctx.noise_shape = None if noise is None else tuple(noise.shape)
        ctx.shapes = [tuple(t.shape) for t in tensors]
        return obs._value()

    @staticmethod
    def backward(ctx, g):
        obs = ctx.obs
        noise_grad, grads = obs.gradients()
        gval = float(g)
        out_noise = None
        if ctx.noise_shape is not None:
            ng = noise_grad * gval
            numel = int(np.prod(ctx.noise_shape)) if ctx.noise_shape else 1
            out_noise = ng.sum().reshape(ctx.noise_shape) if numel == 1 else ng.reshape(ctx.noise_shape)
            out_noise = out_noise.to(obs._noise_device)
        outs = []
        for (kind, ti, fi, tensor), shape in zip(obs._params, ctx.shapes):
            if kind == "coef":
                val = np.asarray(grads["coef"][ti])
            else:
                val = np.asarray(grads["factors"][ti][fi][kind], dtype=np.float64)
            numel = int(np.prod(shape)) if shape else 1
            if val.size != numel:
                val = val.sum()  # a scalar parameter broadcast over several features
            outs.append(torch.as_tensor(np.asarray(val, dtype=np.float64) * gval, dtype=tensor.dtype).reshape(shape).to(tensor.device))
        return (None, out_noise, *outs)


class Measure:
    """Stand-in for stheno's `Measure`: only `logpdf` is used by GPAR (model.py:226)."""

    def logpdf(self, *args):
        if len(args) == 1:
            return args[0].logpdf()
        fdd, y = args
        return fdd.logpdf(y)


class GP:
    """Zero-mean Gaussian process with a `gpar_amd.kernels.Kernel`; `f | obs` gives the posterior process."""

    def __init__(self, kernel, measure=None, engine=None, _obs=None):
        if not isinstance(kernel, Kernel):
            raise TypeError("kernel must be a gpar_amd.kernels.Kernel")
        self.kernel = kernel
        self.measure = measure if measure is not None else Measure()
        self._engine = engine
        self._obs = _obs  # None: prior process

    @property
    def engine(self):
        return self._engine if self._engine is not None else get_engine()

    @property
    def is_posterior(self):
        return self._obs is not None

    def __call__(self, x, noise=None):
        return FDD(self, x, noise)

    def __or__(self, obs):
        if isinstance(obs, tuple):
            obs = Obs(*obs)
        if not isinstance(obs, (Obs, PseudoObs)):
            raise TypeError("can only condition on Obs / PseudoObs or a (fdd, y) tuple")
        if self.is_posterior:
            obs = self._obs.merged_with(obs)
        if obs.prior_gp().kernel is not self.kernel:
            raise ValueError("observations belong to a different process")
        return GP(self.kernel, measure=Measure(), engine=self._engine, _obs=obs)

    def mean(self, x):
        eng = self.engine
        x = _as_matrix(eng, x)
        if not self.is_posterior or x.shape[0] == 0:
            return torch.zeros(x.shape[0], 1, dtype=torch.float64, device=x.device)
        return self._obs.posterior_mean(x)

    def sample_batch(self, xs, noise=None):
        """One draw of f (+ noise) at each input set in `xs` (a list of n* x D matrices, e.g. the per-sample design
        matrices of ancestral sampling): returns an n* x len(xs) matrix.  For a dense posterior the expensive step -
        V_s = K(x_s, X) L^-T for every s - is ONE stacked triangular solve instead of len(xs) separate ones."""
        if self.is_posterior and isinstance(self._obs, Obs):
            return self._obs.posterior_sample_batch(self, xs, noise)
        return torch.cat([self(x_s, noise).sample() for x_s in xs], dim=1)

    def mean_batch(self, xs):
        if self.is_posterior and isinstance(self._obs, Obs) and len(xs) > 1:
            return self._obs.posterior_mean_batch(xs)
        return [self.mean(x_s) for x_s in xs]

    # convenience used by tests / examples
    def marginals(self, x):
        return self(x).marginals()


class FDD:
    """`f(x, noise)`: the process at the rows of x, plus independent noise (scalar or per-row vector)."""

    def __init__(self, p, x, noise=None):
        self.p = p
        self.eng = p.engine
        self.x = _as_matrix(self.eng, x)
        self.n = self.x.shape[0]
        self.noise = _noise_vector(self.eng, noise, self.n)
        self.noise_arg = noise  # kept for the autograd path (may carry a graph)
        self._ck = None
        self._z = None

    # compiled kernel + features of x (cached)
    def features(self):
        if self._z is None:
            self._ck = self.eng.compile(self.p.kernel, self.x.shape[1])
            self._z = self.eng.features(self._ck, self.x)
        return self._ck, self._z

    # ---- moments ---------------------------------------------------------------------------------
    def mean(self):
        return self.p.mean(self.x)

    def _fill_cov(self, block, jitter):
        """Write the lower triangle of cov(f(x)) + diag(noise) + jitter I into `block`; returns the mean (n x 1)."""
        if self.p.is_posterior:
            return self.p._obs.posterior_moments(self, block, jitter)
        ck, z = self.features()
        self.eng.gram(ck, z, lower=True, diag_add=self.noise, diag_const=jitter, out=block)
        return torch.zeros(self.n, 1, dtype=torch.float64, device=self.x.device)

    def var(self):
        """Dense covariance (including noise), symmetric; for tests and small problems."""
        block = self.eng.new_matrix(self.n, self.n)
        self._fill_cov(block, 0.0)
        low = torch.tril(block)
        return low + torch.tril(low, -1).T

    def marginals(self):
        v = self.var()
        return self.mean().reshape(-1), torch.diagonal(v).clone()

    # ---- logpdf / sample ---------------------------------------------------------------------------
    def _factor(self, y):
        eng = self.eng
        y = _as_matrix(eng, y)
        if y.shape[0] != self.n:
            raise ValueError(f"{y.shape[0]} observations for {self.n} inputs")
        holder = {}

        def fill(block):
            holder["mean"] = self._fill_cov(block, eng.epsilon)

        if self.p.is_posterior:
            # the residual needs the mean, which _fill_cov produces: fill into a scratch block first
            scratch = eng.new_matrix(self.n, self.n)
            mean = self._fill_cov(scratch, eng.epsilon)
            resid = y - mean
            return _Factor(eng, self.n, lambda block: block.copy_(scratch), resid)
        return _Factor(eng, self.n, fill, y)

    def logpdf(self, y):
        return self._factor(y).logpdf()

    def sample(self, num=1):
        """num draws, one per column: mean + chol(cov + eps I) randn."""
        eng = self.eng
        n = self.n
        if n == 0:
            return torch.zeros(0, num, dtype=torch.float64, device=self.x.device)
        S = eng.new_matrix(n, n)
        mean = self._fill_cov(S, eng.epsilon)
        _, info = eng.potrf_(S)
        eng.check_info(info)
        zr = eng.randn(n, num)
        out = eng.trmv_lower(S, zr) if num == 1 else eng.gemm(S, zr, a_lower=True)
        return out + mean


class Obs:
    """Exact observations `y = f(x) + noise` (stheno `Obs(fdd, y)`)."""

    def __init__(self, fdd, y):
        self.fdd = fdd
        self.eng = fdd.eng
        self.y = _as_matrix(self.eng, y)
        if self.y.shape[0] != fdd.n:
            raise ValueError(f"{self.y.shape[0]} observations for {fdd.n} inputs")
        self._fac = None

    def prior_gp(self):
        p = self.fdd.p
        return p if not p.is_posterior else p._obs.prior_gp()

    # log marginal likelihood of y under the process the fdd belongs to (prior or posterior)
    def logpdf(self):
        if self.fdd.p.is_posterior:
            return self.fdd.logpdf(self.y)
        if torch.is_grad_enabled():
            params = kernel_parameters(self.fdd.p.kernel)
            noise = self.fdd.noise_arg if _needs_grad(self.fdd.noise_arg) else None
            if params or noise is not None:
                self._params = params
                self._noise_device = None if noise is None else noise.device
                return _LogMarginal.apply(self, noise, *[p[3] for p in params])
        return self.factor().logpdf()

    def _value(self):
        return self.factor().logpdf()

    def gradients(self):
        """(1/2 diag(W) as a device vector, kernel-parameter gradients) with W = alpha alpha^T - (K + D)^-1."""
        eng, fac, n = self.eng, self.factor(), self.fdd.n
        W = eng.chol_inverse(fac.L)  # (K + D)^-1, lower triangle
        a = fac.alpha()
        eng.gemm(a, a, ta=True, alpha=1.0, beta=-1.0, out=W, c_lower=True)
        ck, _ = self.fdd.features()
        grads = eng.kernel_grads(ck, self.fdd.x, W)
        return 0.5 * torch.diagonal(W).clone(), grads

    def adopt_factor(self):
        """Install an empty factor whose buffer the caller fills (a factor computed by another rank)."""
        self._fac = _Factor.placeholder(self.eng, self.fdd.n)
        return self._fac

    def factor(self):
        """Cholesky of K + D + eps I and L^-1 y; valid when the fdd belongs to the prior."""
        if self._fac is None:
            if self.fdd.p.is_posterior:
                raise RuntimeError("internal: factor() requested for observations of a posterior process")
            self._fac = self.fdd._factor(self.y)
        return self._fac

    def merged_with(self, other):
        """Observations of the prior equivalent to conditioning on `self` and then on `other` (dense only)."""
        if not isinstance(other, Obs):
            raise NotImplementedError("conditioning a posterior on inducing-point observations is not supported")
        prior = self.prior_gp()
        a, b = self.fdd, other.fdd
        if a.noise is None or b.noise is None:
            na = a.noise if a.noise is not None else torch.zeros(a.n, dtype=torch.float64, device=a.x.device)
            nb = b.noise if b.noise is not None else torch.zeros(b.n, dtype=torch.float64, device=b.x.device)
        else:
            na, nb = a.noise, b.noise
        x = torch.cat([a.x, b.x], dim=0)
        return Obs(FDD(prior, x, torch.cat([na, nb])), torch.cat([self.y, other.y], dim=0))

    # ---- posterior ---------------------------------------------------------------------------------
    def _cross(self, fdd_or_x):
        """K(x*, X) as an n* x n matrix plus the features of x*."""
        ck, z = self.fdd.features()
        if isinstance(fdd_or_x, FDD):
            zs = self.eng.features(ck, fdd_or_x.x)
        else:
            zs = self.eng.features(ck, fdd_or_x)
        return ck, zs, self.eng.gram(ck, zs, z)

    def posterior_mean(self, x):
        fac = self.factor()
        if self.fdd.n == 0:
            return torch.zeros(x.shape[0], 1, dtype=torch.float64, device=x.device)
        _, _, Ks = self._cross(x)
        return self.eng.gemm(Ks, fac.alpha(), tb=True)

    def posterior_mean_batch(self, xs):
        """Posterior means at several input sets with one stacked cross-Gram product."""
        eng, fac = self.eng, self.factor()
        ck, z = self.fdd.features()
        sizes = [int(x_s.shape[0]) for x_s in xs]
        if self.fdd.n == 0:
            return [torch.zeros(k, 1, dtype=torch.float64, device=z.device) for k in sizes]
        B = eng.new_matrix(sum(sizes), self.fdd.n)
        r = 0
        for x_s, k in zip(xs, sizes):
            eng.gram(ck, eng.features(ck, _as_matrix(eng, x_s)), z, out=B[r : r + k])
            r += k
        means = eng.gemm(B, fac.alpha(), tb=True)
        return list(torch.split(means, sizes, dim=0))

    def posterior_sample_batch(self, gp, xs, noise):
        eng, fac = self.eng, self.factor()
        ck, z = self.fdd.features()
        n, S = self.fdd.n, len(xs)
        ns = int(xs[0].shape[0])
        out = torch.empty(ns, S, dtype=torch.float64, device=z.device)
        if ns == 0:
            return out
        noise_vec = _noise_vector(eng, noise, ns)
        zr = eng.randn(ns, S)
        # bound the stacked right-hand side (S n* x n doubles) to ~16 GB of the 288 GB HBM
        chunk = max(1, int(16e9 // max(1, ns * max(n, 1) * 8)))
        for s0 in range(0, S, chunk):
            s1 = min(S, s0 + chunk)
            zss = [eng.features(ck, _as_matrix(eng, xs[s])) for s in range(s0, s1)]
            B = eng.new_matrix((s1 - s0) * ns, max(n, 1))
            if n > 0:
                for k, zs in enumerate(zss):
                    eng.gram(ck, zs, z, out=B[k * ns : (k + 1) * ns])
                eng.trsm_rlt_(fac.L, B)  # every V_s = K(x_s, X) L^-T in one solve
                means = eng.gemm(B, fac.zrow, tb=True)
            # the per-sample blocks (Gram, SYRK downdate, an n* x n* factorisation, one matvec) are independent: a small
            # factorisation is a latency-bound chain, so they are dealt over a few streams and checked once at the end
            pipe = eng.pipeline(min(4, s1 - s0)) if s1 - s0 > 1 else None
            with eng.defer_checks(), joining(pipe):  # streams are joined BEFORE the deferred info words are read
                for k, zs in enumerate(zss):
                    with (pipe.stage(k, B, zr, out, zs, noise_vec, means if n > 0 else None) if pipe is not None else contextlib.nullcontext()):
                        cov = eng.new_matrix(ns, ns)
                        eng.gram(ck, zs, lower=True, diag_add=noise_vec, diag_const=eng.epsilon, out=cov)
                        mean = 0.0
                        if n > 0:
                            V = B[k * ns : (k + 1) * ns]
                            eng.gemm(V, V, tb=True, alpha=-1.0, beta=1.0, out=cov, c_lower=True)
                            mean = means[k * ns : (k + 1) * ns]
                        _, info = eng.potrf_(cov)
                        eng.check_info(info)
                        out[:, s0 + k : s0 + k + 1] = eng.trmv_lower(cov, zr[:, s0 + k : s0 + k + 1]) + mean
        return out

    def posterior_moments(self, fdd, block, jitter):
        """Lower triangle of K_** - V V^T + diag(noise*) + jitter I into `block`; returns the mean."""
        eng = self.eng
        fac = self.factor()
        ck, zs, V = self._cross(fdd)
        eng.gram(ck, zs, lower=True, diag_add=fdd.noise, diag_const=jitter, out=block)
        if self.fdd.n == 0:
            return torch.zeros(fdd.n, 1, dtype=torch.float64, device=block.device)
        eng.trsm_rlt_(fac.L, V)  # V = K_*x L^-T
        eng.gemm(V, V, tb=True, alpha=-1.0, beta=1.0, out=block, c_lower=True)
        return eng.gemm(V, fac.zrow, tb=True)


class PseudoObs:
    """Inducing-point observations, VFE approximation (stheno `PseudoObs(f(x_ind), f(x, noise), y)`)."""

    def __init__(self, u, fdd, y):
        if isinstance(u, tuple):
            raise TypeError("pass a single FDD of inducing points")
        self.u = u
        self.fdd = fdd
        self.eng = fdd.eng
        self.y = _as_matrix(self.eng, y)
        if fdd.p.is_posterior or u.p.is_posterior:
            raise NotImplementedError("inducing-point observations must be built on the prior")
        if fdd.noise is None:
            raise ValueError("inducing-point observations need observation noise")
        self._state = None

    def prior_gp(self):
        return self.fdd.p

    def merged_with(self, other):
        raise NotImplementedError("conditioning a sparse posterior again is not supported")

    def _compute(self):
        if self._state is not None:
            return self._state
        eng = self.eng
        n, M = self.fdd.n, self.u.n
        ck, zx = self.fdd.features()
        zu = eng.features(ck, self.u.x)
        d = self.fdd.noise
        # L_z = chol(K_zz + eps I)
        Kzz = eng.new_matrix(M, M)
        eng.gram(ck, zu, lower=True, diag_const=eng.epsilon, out=Kzz)
        _, info = eng.potrf_(Kzz)
        eng.check_info(info)
        Lz = Kzz
        # B^T = K_xz L_z^-T  (n x M), rows scaled by d^-1/2
        Bt = eng.gram(ck, zx, zu)
        eng.trsm_rlt_(Lz, Bt)
        kdiag = eng.gram_diag(ck, zx)
        trace_term = torch.sum((kdiag - torch.sum(Bt * Bt, dim=1)) / d)
        rs = torch.rsqrt(d)
        # [B D^-1/2 | D^-1/2 y]: ONE product over the n data points gives A - I, c = B D^-1 y and y^T D^-1 y together
        Bs = eng.new_matrix(n, M + 1)
        torch.mul(Bt, rs[:, None], out=Bs[:, :M])
        Bs[:, M] = self.y.reshape(-1) * rs
        G = eng.gemm(Bs, Bs, ta=True, c_lower=True)  # (M + 1) x (M + 1), lower triangle
        c = G[M : M + 1, :M].clone()
        yDy = G[M, M].clone()

        def fill(block):
            block.copy_(G[:M, :M])
            block.diagonal().add_(1.0)

        facA = _Factor(eng, M, fill, c)
        elbo = -0.5 * (trace_term + torch.sum(torch.log(d)) + n * _LOG_2PI + facA.logdet[0] + yDy - facA.quad)
        # v = L_z^-T A^-1 c, so that mean(x*) = K_*z v
        v = facA.alpha().clone()  # (A^-1 c)^T, 1 x M
        eng.trsm_rln_(Lz, v)
        deferring = getattr(eng, "_deferred", None) is not None
        self._state = {"ck": ck, "zu": zu, "Lz": Lz, "La": facA.L, "v": v, "elbo": elbo.detach() if deferring else elbo.detach().cpu(),
                       "Bt": Bt, "facA": facA, "kdiag": kdiag}
        return self._state

    def _value(self):
        return self._compute()["elbo"]

    def logpdf(self):
        """The VFE bound; differentiable with respect to kernel parameters and noise when they carry a graph."""
        if torch.is_grad_enabled():
            params = kernel_parameters(self.fdd.p.kernel)
            noise = self.fdd.noise_arg if _needs_grad(self.fdd.noise_arg) else None
            if params or noise is not None:
                self._params = params
                self._noise_device = None if noise is None else noise.device
                return _LogMarginal.apply(self, noise, *[p[3] for p in params])
        return self._value()

    elbo = logpdf

    def gradients(self):
        """(dF/dd as a device vector, kernel-parameter gradients) of the VFE bound F (Titsias 2009, eq. 9).

        With S = Q + D, Q = K_fu K_uu^-1 K_uf, G = alpha alpha^T - S^-1 (alpha = S^-1 y) and P = K_uu^-1 K_uf:
            dF = sum_aj [W_fu]_aj dK_fu[a, j] + sum_ij [W_uu]_ij dK_uu[i, j] - 1/2 sum_a dk_aa / d_a + sum_a g_a dd_a,
            W_fu = (G + D^-1) P^T,   W_uu = -1/2 P (G + D^-1) P^T,   g_a = 1/2 (G_aa + (k_aa - q_aa) / d_a^2).
        Nothing n x n is formed: with B = L_z^-1 K_uf, A = I + B D^-1 B^T (both available from the forward pass),
            G + D^-1 = alpha alpha^T + D^-1 B^T A^-1 B D^-1,
            W_fu = [alpha beta^T + D^-1 B^T (I - A^-1)] L_z^-1,   beta = B alpha,
            W_uu = -1/2 L_z^-T (beta beta^T + A - 2 I + A^-1) L_z^-1,
            (S^-1)_aa = 1/d_a - |L_A^-1 B_:a|^2 / d_a^2.
        The three weighted sums over kernel derivatives are one fused device pass each (`kernel_grads_vfe`)."""
        eng = self.eng
        st = self._compute()
        n, M = self.fdd.n, self.u.n
        d = self.fdd.noise
        Bt, facA, Lz = st["Bt"], st["facA"], st["Lz"]
        a = facA.alpha()  # (A^-1 B D^-1 y)^T, 1 x M
        alpha = (self.y.reshape(-1) - eng.gemm(Bt, a, tb=True).reshape(-1)) / d  # S^-1 y
        beta = eng.gemm(alpha.reshape(1, n), Bt)  # 1 x M
        Ainv = eng.chol_inverse(facA.L)  # lower triangle of A^-1
        Ainv_full = torch.tril(Ainv) + torch.tril(Ainv, -1).T
        # W_fu
        T = eng.new_matrix(n, M)
        T.copy_(Bt)
        eng.gemm(Bt, Ainv_full, alpha=-1.0, beta=1.0, out=T)  # B^T (I - A^-1)
        T.div_(d[:, None])
        T.add_(alpha[:, None] * beta.reshape(1, M))
        eng.trsm_rln_(Lz, T)  # ... L_z^-1
        # W_uu
        S = eng.new_matrix(M, M)
        S.copy_(beta.reshape(M, 1) * beta.reshape(1, M) + Ainv_full)
        A_minus_I = eng.gemm(Bt / torch.sqrt(d)[:, None], Bt / torch.sqrt(d)[:, None], ta=True)  # B D^-1 B^T
        S.add_(A_minus_I)
        S.diagonal().sub_(1.0)
        eng.trsm_rln_(Lz, S)  # S L_z^-1
        St = eng.new_matrix(M, M)
        St.copy_(S.T)
        eng.trsm_rln_(Lz, St)  # L_z^-T S L_z^-1 (symmetric)
        Wuu = eng.new_matrix(M, M)
        Wuu.copy_(-0.25 * (St + St.T))
        # per-point terms
        E = eng.new_matrix(n, M)
        E.copy_(Bt)
        eng.trsm_rlt_(facA.L, E)  # rows: (L_A^-1 B_:a)^T
        e = torch.sum(E * E, dim=1)
        q = torch.sum(Bt * Bt, dim=1)
        noise_grad = 0.5 * (alpha * alpha - 1.0 / d + e / (d * d) + (st["kdiag"] - q) / (d * d))
        grads = eng.kernel_grads_vfe(st["ck"], self.fdd.x, self.u.x, T, Wuu, -0.5 / d)
        return noise_grad, grads

    def posterior_mean(self, x):
        st = self._compute()
        zs = self.eng.features(st["ck"], x)
        Ksz = self.eng.gram(st["ck"], zs, st["zu"])
        return self.eng.gemm(Ksz, st["v"], tb=True)

    def posterior_moments(self, fdd, block, jitter):
        eng = self.eng
        st = self._compute()
        ck = st["ck"]
        zs = eng.features(ck, fdd.x)
        eng.gram(ck, zs, lower=True, diag_add=fdd.noise, diag_const=jitter, out=block)
        P = eng.gram(ck, zs, st["zu"])
        mean = eng.gemm(P, st["v"], tb=True)
        eng.trsm_rlt_(st["Lz"], P)  # P = K_*z L_z^-T
        eng.gemm(P, P, tb=True, alpha=-1.0, beta=1.0, out=block, c_lower=True)
        eng.trsm_rlt_(st["La"], P)  # Q = P L_A^-T
        eng.gemm(P, P, tb=True, alpha=1.0, beta=1.0, out=block, c_lower=True)
        return mean


SparseObs = PseudoObs
