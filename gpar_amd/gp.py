"""GP layer objects: the stheno subset GPAR drives, computed on the MI355X through the engine seam.

The reference treats a layer as `(GP, noise)` and uses exactly these operations on it
(/root/reference/gpar/model.py): `f(x, noise)` -> finite-dimensional distribution (:264,270,287,289),
`Obs(fdd, y)` / `PseudoObs(f(x_ind), fdd, y)` (:286-289), `f.measure.logpdf(obs)` (:226), `f | obs` (:170,232,298),
`.mean(x)` (:299,301), `fdd.sample()` (:235,264,270) and, in its tests, `fdd.logpdf(y)` and `f | (fdd, y)`.
This module provides those names with the same call signatures; the arithmetic is

  exact GP (Rasmussen & Williams alg. 2.1), evaluated with ONE partial Cholesky of an augmented matrix

        [ K + D + eps I      .  ]   factor the first n columns     [ L          . ]
        [ r^T                0  ]   ------------------------>      [ (L^-1 r)^T   -|L^-1 r|^2 ]

    which yields log|K|, the quadratic form and L in a single device pipeline (include/gpar_hip.h:
    gpar_potrf); posterior moments at new inputs are V = K_*x L^-T (TRSM), mean = V (L^-1 r),
    cov = K_** - V V^T (SYRK);

  inducing points (Titsias 2009, VFE; stheno's PseudoObs default):  L_z = chol(K_zz), B^T = K_xz L_z^-T,
    A = I + B D^-1 B^T, bound = -1/2 [ sum_j (k_jj - |B_:j|^2)/d_j + sum_j log(2 pi d_j) + log|A| + r^T D^-1 r - |L_A^-1 B D^-1 r|^2 ].

A process is either the prior (zero mean, kernel k) or a posterior `g | obs`, whose mean and kernel are those of the
process the observations were made of, corrected by the observations:

    dense obs   mean + V_a (L^-1 r),        k(a, b) - V_a V_b^T,                 V_a = k(a, X) L^-T
    sparse obs  mean + k(a, Z) v,           k(a, b) - P_a P_b^T + Q_a Q_b^T,     P_a = k(a, Z) L_z^-T, Q_a = P_a L_A^-T

so observations of a posterior (a posterior log-density; conditioning twice) work for every combination of dense and
inducing-point observations by recursion over the chain (`GP._moments_into / _cross / _diag / _mean_at`); the common
cases - observations of the prior, dense observations of a dense posterior - keep their one-factorisation fast paths.

All heavy steps are engine primitives (HIP kernels); torch is used here only for O(n) vector plumbing.
"""
import contextlib
import os
import math

import numpy as np
import torch

from .engine import get_engine, joining
from .kernels import Kernel

__all__ = ["Measure", "GP", "FDD", "Obs", "PseudoObs", "PseudoObsVFE", "PseudoObsFITC", "PseudoObsDTC", "SparseObs"]

_LOG_2PI = math.log(2.0 * math.pi)


def one_call_grad_rows():
    """Layers with at most this many rows evaluate their training objective and its gradient through ONE library call
    (gpar_logpdf_dense_grad; GPAR_ONE_CALL_GRAD_ROWS, 0 = never): the same launches, without ~30 host-side steps and two of the
    three synchronisations around them.  Above, the host's share of an evaluation is negligible and the two-step form lets the
    factor be reused."""
    return int(os.environ.get("GPAR_ONE_CALL_GRAD_ROWS", "4096"))


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


def _zeros(n, like):
    return torch.zeros(n, 1, dtype=torch.float64, device=like.device)


class _Pts:
    """A set of inputs together with the compiled kernel and their features under it (shared by every process of one
    conditioning chain: conditioning changes the mean and the kernel's correction terms, not the base kernel)."""

    __slots__ = ("x", "ck", "z", "n")

    def __init__(self, eng, kernel, x):
        self.x = x
        self.n = int(x.shape[0])
        self.ck = eng.compile(kernel, x.shape[1])
        self.z = eng.features(self.ck, x)


class _Factor:
    """Cholesky of an n x n SPD matrix together with the solve against one right-hand side, from one partial
    factorisation of the (n + 1) x (n + 1) augmented matrix.  `fill(block)` must write the lower triangle of the
    SPD matrix into `block` (an n x n view) and return the vector to subtract from `rhs` (or None)."""

    def __init__(self, eng, n, fill, rhs):
        self.eng, self.n = eng, n
        A = eng.new_matrix(n + 1, n + 1)
        logdet = torch.zeros(1, dtype=torch.float64, device=A.device)
        scale = 1.0
        while n > 0:
            shift = fill(A[:n, :n], scale)
            row = rhs.reshape(-1)
            A[n, :n] = row if shift is None else row - shift.reshape(-1)
            A[n, n].zero_()  # (NOT `A[n, n] = 0.0`: a Python scalar assigned by index goes through a CPU tensor, a
            #                   synchronous host-to-device copy queued behind everything on the stream - a hidden host sync)
            logdet, info = eng.potrf_(A, nf=n)
            if not _retrying(eng, info, scale):
                break
            scale *= 10.0  # lab's Cholesky retry: jitter x 10 (B.cholesky_retry_factor)
        if n == 0:
            A[n, n].zero_()
        self.A = A
        self.L = A[:n, :n]
        self.zrow = A[n : n + 1, :n]  # (L^-1 rhs)^T, 1 x n
        self.logdet = logdet
        self.quad = -A[n, n]  # |L^-1 rhs|^2 (device scalar)
        self._alpha = None

    @classmethod
    def from_batch(cls, eng, n, A, logdet):
        """The factor of one layer out of a lock-step batch (HipEngine.factor_dense_batch): `A` its (n + 1) x (n + 1) block,
        `logdet` its word."""
        self = cls.__new__(cls)
        self.eng, self.n = eng, n
        self.A = A
        self.L = A[:n, :n]
        self.zrow = A[n : n + 1, :n]
        self.logdet = logdet
        self.quad = -A[n, n]
        self._alpha = None
        return self

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


def _retrying(eng, info, scale):
    """Record / check the verdict of a factorisation.  True: it failed and lab's retry rule (engine.cholesky_retry_factor)
    allows another attempt with ten times the jitter."""
    from .engine import NotPositiveDefiniteError

    limit = getattr(eng, "cholesky_retry_factor", 1.0)
    if limit <= 1.0:
        eng.check_info(info)  # possibly deferred: no host sync
        return False
    try:
        eng._raise_for(info)  # synchronises: a retry needs the verdict now
    except NotPositiveDefiniteError:
        if scale * 10.0 > limit:
            raise
        return True
    return False


@contextlib.contextmanager
def _on_side(stream):
    """Run the block on `stream` after everything enqueued on the current stream so far."""
    stream.wait_stream(torch.cuda.current_stream(stream.device))
    with torch.cuda.stream(stream):
        yield


def _join_side(stream, *tensors):
    """The current stream waits for `stream`; tensors allocated there are handed over to the current stream."""
    main = torch.cuda.current_stream(stream.device)
    main.wait_stream(stream)
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            t.record_stream(main)


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


def _param_grads(params, shapes, grads, scale):
    """Gradient dictionary of an engine pass (coef / per-factor scales, periods, alpha) -> one tensor per kernel parameter."""
    outs = []
    for (kind, ti, fi, tensor), shape in zip(params, shapes):
        if kind == "coef":
            val = np.asarray(grads["coef"][ti])
        else:
            val = np.asarray(grads["factors"][ti][fi][kind], dtype=np.float64)
        numel = int(np.prod(shape)) if shape else 1
        if val.size != numel:
            val = val.sum()  # a scalar parameter broadcast over several features
        outs.append(torch.as_tensor(np.asarray(val, dtype=np.float64) * scale, dtype=tensor.dtype).reshape(shape).to(tensor.device))
    return outs


def _add_grads(a, b):
    """Sum of two engine gradient dictionaries of the same kernel."""
    if a is None:
        return b
    out = {"coef": [x + y for x, y in zip(a["coef"], b["coef"])], "factors": []}
    for fa, fb in zip(a["factors"], b["factors"]):
        out["factors"].append([{k: (None if ga[k] is None else np.asarray(ga[k]) + np.asarray(gb[k])) for k in ga} for ga, gb in zip(fa, fb)])
    return out


def _shaped_noise_grad(noise_grad, shape, device):
    numel = int(np.prod(shape)) if shape else 1
    out = noise_grad.sum().reshape(shape) if numel == 1 else noise_grad.reshape(shape)
    return out.to(device)


class _LogMarginal(torch.autograd.Function):
    """log N(y; 0, K_theta(X) + D) - or, for inducing-point observations, the VFE / DTC bound - as a differentiable
    function of the kernel parameters, the noise vector and (when they are themselves functions of hyper-parameters: fed
    forward posterior means, or inducing inputs being optimised) the inputs X and inducing inputs Z.

    Forward is the fused Gram + augmented Cholesky on the device.  Backward is analytic:
    with W = alpha alpha^T - K^-1,  d/dtheta = 1/2 sum_ab W_ab dK_ab/dtheta and d/dX = 1/2 d/dX sum_ab W_ab k(x_a, x_b);
    K^-1 comes from L (TRSM on the identity + SYRK), the kernel-parameter sums from one fused pass over W, the input
    gradients from a second one (csrc/gram.h).  torch then chains these through whatever produced the arguments (bound
    transforms, noise / w, `_update_inputs`' concatenations, `_PosteriorMean`).
    """

    @staticmethod
    def forward(ctx, obs, noise, X, Z, *tensors):
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
            out_noise = _shaped_noise_grad(noise_grad * gval, ctx.noise_shape, obs._noise_device)
        gX = gZ = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            gX, gZ = obs.input_gradients(ctx.needs_input_grad[2], ctx.needs_input_grad[3])
            gX = None if gX is None else gX * gval
            gZ = None if gZ is None else gZ * gval
        # the weights of the backward pass (n x n for dense observations: 2 GB at n = 16384) are not needed again
        for name in ("_W", "_weights"):
            if hasattr(obs, name):
                delattr(obs, name)
        return (None, out_noise, gX, gZ, *_param_grads(obs._params, ctx.shapes, grads, gval))


class _PosteriorMean(torch.autograd.Function):
    """Posterior mean of `f | obs` at x* as a differentiable function of the kernel parameters, the noise, the training
    inputs X, the inducing inputs Z and x* itself: the column `_update_inputs` feeds to the next layer
    (reference gpar/model.py:291-322, differentiated by torch autograd there when fit(fix=False) trains layers jointly).
    Backward: `Obs.mean_gradients` / `PseudoObs.mean_gradients` (weighted kernel-derivative passes on the device)."""

    @staticmethod
    def forward(ctx, obs, xs_value, xs, noise, X, Z, *tensors):
        ctx.obs, ctx.xs = obs, xs_value
        ctx.noise_shape = None if noise is None else tuple(noise.shape)
        ctx.noise_device = None if noise is None else noise.device
        ctx.shapes = [tuple(t.shape) for t in tensors]
        ctx.params = kernel_parameters(obs.base.kernel)
        return obs.mean_at(obs.base._pts(xs_value))

    @staticmethod
    def backward(ctx, g):
        obs = ctx.obs
        need = ctx.needs_input_grad
        out = obs.mean_gradients(ctx.xs, g.detach().reshape(-1), want_xs=need[2], want_x=need[4], want_z=need[5])
        g_noise = None if ctx.noise_shape is None else _shaped_noise_grad(out["noise"], ctx.noise_shape, ctx.noise_device)
        return (None, None, out.get("xs"), g_noise, out.get("x"), out.get("z"), *_param_grads(ctx.params, ctx.shapes, out["params"], 1.0))


class Measure:
    """Stand-in for stheno's `Measure`: only `logpdf` is used by GPAR (model.py:226)."""

    def logpdf(self, *args):
        if len(args) == 1:
            return args[0].logpdf()
        fdd, y = args
        return fdd.logpdf(y)


class GP:
    """Gaussian process with a `gpar_amd.kernels.Kernel`: the zero-mean prior, or (after `f | obs`) a posterior."""

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
        if obs.base.kernel is not self.kernel:
            raise ValueError("observations belong to a different process")
        if obs.base is not self:
            if not (obs.base.is_posterior or self.is_posterior):
                obs = obs.rebased(self)  # another handle of the same prior
            else:
                raise ValueError("observations must be made of the process that is being conditioned")
        if self.is_posterior and isinstance(obs, Obs) and isinstance(self._obs, Obs) and not self._obs.base.is_posterior:
            # dense on dense: one factorisation of the stacked observations of the prior
            return GP(self.kernel, measure=Measure(), engine=self._engine, _obs=self._obs.merged_with(obs))
        return GP(self.kernel, measure=Measure(), engine=self._engine, _obs=obs)

    # ---- process primitives (recursive over the conditioning chain) ---------------------------------
    def _pts(self, x):
        eng = self.engine
        return _Pts(eng, self.kernel, _as_matrix(eng, x))

    def _mean_at(self, p):
        if not self.is_posterior or p.n == 0:
            return _zeros(p.n, p.x)
        return self._obs.mean_at(p)

    def _cross(self, pa, pb, row_scale=None):
        """k(a, b) as a new na x nb matrix (rows optionally scaled)."""
        if not self.is_posterior:
            return self.engine.gram(pa.ck, pa.z, pb.z, row_scale=row_scale)
        K = self._obs.cross(pa, pb)
        if row_scale is not None:
            K.mul_(row_scale[:, None])
        return K

    def _diag(self, p):
        """k(a, a) for every row of p (vector)."""
        if not self.is_posterior:
            return self.engine.gram_diag(p.ck, p.z)
        return self._obs.diag(p)

    def _moments_into(self, p, block, diag_add=None, jitter=0.0):
        """Lower triangle of k(p, p) + diag(diag_add) + jitter I into `block`; returns the mean at p (n x 1)."""
        if not self.is_posterior:
            self.engine.gram(p.ck, p.z, lower=True, diag_add=diag_add, diag_const=jitter, out=block)
            return _zeros(p.n, p.x)
        return self._obs.moments_into(p, block, diag_add, jitter)

    # ---- public ------------------------------------------------------------------------------------
    def mean(self, x):
        eng = self.engine
        xm = _as_matrix(eng, x)
        if self.is_posterior and torch.is_grad_enabled() and not self._obs.base.is_posterior and xm.shape[0] > 0 and self._obs.fdd.n > 0:
            obs = self._obs
            params = kernel_parameters(self.kernel)
            noise = obs.fdd.noise_arg if _needs_grad(obs.fdd.noise_arg) else None
            X = obs.fdd.x if obs.fdd.x.requires_grad else None
            Z = obs.u.x if isinstance(obs, PseudoObs) and obs.u.x.requires_grad else None
            xs = xm if xm.requires_grad else None
            if params or noise is not None or X is not None or Z is not None or xs is not None:
                return _PosteriorMean.apply(obs, xm.detach(), xs, noise, X, Z, *[p[3] for p in params])
        return self._mean_at(self._pts(xm))

    def marginal_moments(self, x, noise=None):
        """(mean, variance) of f(x) (+ noise) point by point - no n* x n* covariance is formed."""
        p = self._pts(x)
        var = self._diag(p)
        nv = _noise_vector(self.engine, noise, p.n)
        if nv is not None:
            var = var + nv
        return self._mean_at(p), var

    def sample_batch(self, xs, noise=None):
        """One draw of f (+ noise) at each input set in `xs` (a list of n* x D matrices, e.g. the per-sample design
        matrices of ancestral sampling): returns an n* x len(xs) matrix.  For a dense posterior the expensive step -
        V_s = K(x_s, X) L^-T for every s - is ONE stacked triangular solve instead of len(xs) separate ones."""
        if self.is_posterior and (self._obs.fast_dense or getattr(self._obs, "fast_sparse", False)):
            return self._obs.posterior_sample_batch(xs, noise)
        return torch.cat([self(x_s, noise).sample() for x_s in xs], dim=1)

    def marginal_sample_batch(self, xs, noise=None):
        """As `sample_batch`, but every point is drawn from its own marginal N(mean_j, var_j + noise_j) instead of the
        joint law of the n* points: per-point predictive statistics are unchanged, the draws of one sample are
        independent across points.  Needs no n* x n* covariance and no factorisation of one."""
        eng = self.engine
        S = len(xs)
        ns = int(xs[0].shape[0])
        if self.is_posterior and (self._obs.fast_dense or getattr(self._obs, "fast_sparse", False)):
            mean, var = self._obs.posterior_marginals_batch(xs)
        else:
            moments = [self.marginal_moments(x_s) for x_s in xs]
            mean = torch.cat([m for m, _ in moments], dim=1)
            var = torch.stack([v for _, v in moments], dim=1)
        nv = _noise_vector(eng, noise, ns)
        if nv is not None:
            var = var + nv[:, None]
        # eps as the joint sampler's jitter; a variance rounded below zero is clamped
        return mean + torch.sqrt(torch.clamp(var + eng.epsilon, min=0.0)) * eng.randn(ns, S)

    def mean_batch(self, xs):
        if self.is_posterior and (self._obs.fast_dense or getattr(self._obs, "fast_sparse", False)) and len(xs) > 1:
            return self._obs.posterior_mean_batch(xs)
        return [self.mean(x_s) for x_s in xs]

    # convenience used by tests / examples
    def marginals(self, x):
        return self(x).marginals()


class Stacked:
    """`count` input sets of `rows` points each, stacked by rows in ONE matrix: what ancestral sampling hands from layer to layer
    (each sample's design matrix differs in its last columns).  Behaves like the list of its blocks; the batched posterior
    routines take the matrix as it is (one featurize / cross-Gram launch) instead of concatenating a list again."""

    def __init__(self, matrix, count, shared_cols=0):
        self.matrix, self.count = matrix, int(count)
        self.rows = int(matrix.shape[0]) // max(self.count, 1)
        self.shared_cols = int(shared_cols)   # the leading columns that are the same in every set (Stacked.repeat: all of x)

    @classmethod
    def repeat(cls, x, count):
        return cls(x.repeat(count, 1), count, shared_cols=int(x.shape[1]))

    def __len__(self):
        return self.count

    def __getitem__(self, s):
        if isinstance(s, slice):
            a, b, step = s.indices(self.count)
            if step != 1:
                raise IndexError("contiguous slices only")
            return Stacked(self.matrix[a * self.rows : b * self.rows], max(b - a, 0), self.shared_cols)
        if s < 0:
            s += self.count
        return self.matrix[s * self.rows : (s + 1) * self.rows]

    def __iter__(self):
        return (self[s] for s in range(self.count))

    def with_columns(self, cols):
        """A new stack with `cols` appended to every set: rows x count (column s goes to set s) or already stacked
        ((count rows) x k)."""
        if cols.shape[0] == self.rows and cols.shape[0] != self.matrix.shape[0]:
            cols = cols.T.reshape(-1, 1)
        return Stacked(torch.cat([self.matrix, cols], dim=1), self.count, self.shared_cols)


class FDD:
    """`f(x, noise)`: the process at the rows of x, plus independent noise (scalar or per-row vector)."""

    def __init__(self, p, x, noise=None):
        self.p = p
        self.eng = p.engine
        self.x = _as_matrix(self.eng, x)
        self.n = self.x.shape[0]
        self.noise = _noise_vector(self.eng, noise, self.n)
        self.noise_arg = noise  # kept for the autograd path (may carry a graph)
        self._pts = None

    def pts(self):
        """Inputs + compiled kernel + features (cached)."""
        if self._pts is None:
            self._pts = _Pts(self.eng, self.p.kernel, self.x)
        return self._pts

    def features(self):
        p = self.pts()
        return p.ck, p.z

    # ---- moments ---------------------------------------------------------------------------------
    def mean(self):
        return self.p._mean_at(self.pts())

    def _fill_cov(self, block, jitter):
        """Write the lower triangle of cov(f(x)) + diag(noise) + jitter I into `block`; returns the mean (n x 1)."""
        return self.p._moments_into(self.pts(), block, self.noise, jitter)

    def var(self):
        """Dense covariance (including noise), symmetric; for tests and small problems."""
        block = self.eng.new_matrix(self.n, self.n)
        self._fill_cov(block, 0.0)
        low = torch.tril(block)
        return low + torch.tril(low, -1).T

    def marginals(self):
        mean, var = self.p.marginal_moments(self.x, self.noise)
        return mean.reshape(-1), var

    # ---- logpdf / sample ---------------------------------------------------------------------------
    def _factor(self, y):
        eng = self.eng
        y = _as_matrix(eng, y)
        if y.shape[0] != self.n:
            raise ValueError(f"{y.shape[0]} observations for {self.n} inputs")
        return _Factor(eng, self.n, lambda block, scale: self._fill_cov(block, eng.epsilon * scale), y)

    def logpdf(self, y):
        return self._factor(y).logpdf()

    def sample(self, num=1):
        """num draws, one per column: mean + chol(cov + eps I) randn."""
        eng = self.eng
        n = self.n
        if n == 0:
            return torch.zeros(0, num, dtype=torch.float64, device=self.x.device)
        S = eng.new_matrix(n, n)
        scale = 1.0
        while True:
            mean = self._fill_cov(S, eng.epsilon * scale)
            _, info = eng.potrf_(S)
            if not _retrying(eng, info, scale):
                break
            scale *= 10.0
        zr = eng.randn(n, num)
        out = eng.trmv_lower(S, zr) if num == 1 else eng.gemm(S, zr, a_lower=True)
        return out + mean


class Obs:
    """Exact observations `y = f(x) + noise` (stheno `Obs(fdd, y)`) of the process `fdd.p` (prior or posterior)."""

    def __init__(self, fdd, y):
        self.fdd = fdd
        self.eng = fdd.eng
        self.base = fdd.p
        self.y = _as_matrix(self.eng, y)
        if self.y.shape[0] != fdd.n:
            raise ValueError(f"{self.y.shape[0]} observations for {fdd.n} inputs")
        self._fac = None

    @property
    def fast_dense(self):
        """Dense observations of the prior: the stacked / batched posterior routines below apply."""
        return not self.base.is_posterior

    def prior_gp(self):
        p = self.base
        while p.is_posterior:
            p = p._obs.base
        return p

    def rebased(self, gp):
        return Obs(FDD(gp, self.fdd.x, self.fdd.noise_arg), self.y)

    # log marginal likelihood of y under the process the fdd belongs to (prior or posterior)
    def logpdf(self):
        if not self.base.is_posterior and torch.is_grad_enabled():
            params = kernel_parameters(self.base.kernel)
            noise = self.fdd.noise_arg if _needs_grad(self.fdd.noise_arg) else None
            X = self.fdd.x if self.fdd.x.requires_grad else None
            if params or noise is not None or X is not None:
                self._params = params
                self._noise_device = None if noise is None else noise.device
                # value and gradient ingredients in ONE library call when nothing but kernel parameters and noise is differentiated
                self._fuse_grad = X is None and self._fusable_grad()
                return _LogMarginal.apply(self, noise, X, None, *[p[3] for p in params])
        if self._value_only():
            # one library call: features, Gram, observations, factorisation, value (the factor is not kept)
            eng = self.eng
            ck = eng.compile(self.base.kernel, self.fdd.x.shape[1])
            value, info = eng.logpdf_dense(ck, self.fdd.x, self.y, self.fdd.noise, eng.epsilon)
            eng.check_info(info)
            return value.detach()
        return self.factor().logpdf()

    def _value_only(self):
        """The caller wants the number and nothing else (`transient`, set by GPAR.logpdf for layers that feed nobody), of a
        prior process, with checks deferred (the value stays on the device) and no retry ladder to climb."""
        eng = self.eng
        return (getattr(self, "transient", False) and self._fac is None and not self.base.is_posterior and self.fdd.n > 0
                and hasattr(eng, "logpdf_dense") and getattr(eng, "_deferred", None) is not None
                and getattr(eng, "cholesky_retry_factor", 1.0) <= 1.0 and isinstance(self.y, torch.Tensor) and self.y.is_cuda)

    def _fusable_grad(self):
        """The objective-and-gradient entry point applies: dense observations of a prior process, nothing factored yet, checks
        deferred (the value stays on the device), no retry ladder, small enough for the caller's side to matter."""
        eng = self.eng
        return (self._fac is None and not self.base.is_posterior and 0 < self.fdd.n <= one_call_grad_rows() and hasattr(eng, "logpdf_dense_grad")
                and getattr(eng, "_deferred", None) is not None and getattr(eng, "cholesky_retry_factor", 1.0) <= 1.0
                and isinstance(self.y, torch.Tensor) and self.y.is_cuda)

    def _value(self):
        fuse, self._fuse_grad = getattr(self, "_fuse_grad", False), False   # (decided per call by logpdf(): never sticky)
        if fuse and self._fac is None:
            eng = self.eng
            ck = eng.compile(self.base.kernel, self.fdd.x.shape[1])
            value, info, self._fused_gradients, (A, logdet) = eng.logpdf_dense_grad(ck, self.fdd.x.detach(), self.y, self.fdd.noise, eng.epsilon)
            eng.check_info(info)
            # the call leaves the factor behind: a posterior mean / conditioning on these observations (fit(fix=False), replace,
            # imputation) takes it from here instead of factoring the same matrix again
            self._fac = _Factor.from_batch(eng, self.fdd.n, A, logdet)
            return value.detach()
        return self.factor().logpdf()

    def gradients(self):
        """(1/2 diag(W) as a device vector, kernel-parameter gradients) with W = alpha alpha^T - (K + D)^-1."""
        fused = getattr(self, "_fused_gradients", None)
        if fused is not None:
            self._fused_gradients = None
            return fused()
        eng, fac = self.eng, self.factor()
        if hasattr(eng, "chol_inverse_x") and fac.n > 0:
            # alpha = (K + D)^-1 y = X (L^-1 y) with X = L^-T, which the inverse has just formed: a triangular matrix-vector product at
            # memory speed instead of a backward substitution (the launches of gpar_logpdf_dense_grad, same bits)
            W, X = eng.chol_inverse_x(fac.L)  # (K + D)^-1, lower triangle; L^-T, upper triangle
            a = eng.trmv_upper(X, fac.zrow)
            del X
        else:
            W = eng.chol_inverse(fac.L)  # (K + D)^-1, lower triangle
            a = fac.alpha()
        eng.gemm(a, a, ta=True, alpha=1.0, beta=-1.0, out=W, c_lower=True)
        ck, _ = self.fdd.features()
        grads = eng.kernel_grads(ck, self.fdd.x.detach(), W)
        self._W = W
        return 0.5 * torch.diagonal(W).clone(), grads

    def input_gradients(self, want_x, want_z):
        """d logpdf / d X = 1/2 d/dX sum_ab W_ab k(x_a, x_b) (both arguments of k move)."""
        ck, _ = self.fdd.features()
        gX = 0.5 * self.eng.kernel_input_grads(ck, self.fdd.x.detach(), None, self._W, sym=True) if want_x else None
        return gX, None

    def mean_gradients(self, xs, g, want_xs=True, want_x=True, want_z=False):
        """Gradients of  g^T mean(xs),  mean(xs) = K(xs, X) alpha, alpha = S^-1 y, S = K(X, X) + D:
             d = sum_ab (g_a alpha_b) dK(xs_a, X_b) - sum_bc (beta_b alpha_c) dS_bc,      beta = S^-1 K(X, xs) g."""
        eng, fac = self.eng, self.factor()
        ck, z = self.fdd.features()
        X = self.fdd.x.detach()
        n = self.fdd.n
        alpha = fac.alpha().reshape(-1)  # n
        pxs = self.base._pts(xs)
        Ks = eng.gram(ck, pxs.z, z)  # n* x n
        beta = eng.gemv_t(Ks, g).reshape(1, n).clone()
        eng.trsm_rlt_(fac.L, beta)
        eng.trsm_rln_(fac.L, beta)
        beta = beta.reshape(-1)
        W_rect = eng.new_matrix(xs.shape[0], n)
        W_rect.copy_(g[:, None] * alpha[None, :])
        W_sym = eng.new_matrix(n, n)
        W_sym.copy_(-0.5 * (beta[:, None] * alpha[None, :] + alpha[:, None] * beta[None, :]))
        params = _add_grads(eng.kernel_grads_weighted(ck, xs, X, W_rect), eng.kernel_grads_weighted(ck, X, None, W_sym, sym=True))
        out = {"params": params, "noise": -beta * alpha}
        if want_xs:
            out["xs"] = eng.kernel_input_grads(ck, xs, X, W_rect)
        if want_x:
            Wt = eng.new_matrix(n, xs.shape[0])
            Wt.copy_(W_rect.T)
            out["x"] = eng.kernel_input_grads(ck, X, xs, Wt) + eng.kernel_input_grads(ck, X, None, W_sym, sym=True)
        return out

    def adopt_factor(self):
        """Install an empty factor whose buffer the caller fills (a factor computed by another rank)."""
        self._fac = _Factor.placeholder(self.eng, self.fdd.n)
        return self._fac

    def _batchable(self):
        """This observation's factor may come out of a lock-step batch: a prior process (zero mean: the right-hand side is y
        itself), nothing factored yet, checks deferred, no retry ladder to climb."""
        eng = self.eng
        return (self._fac is None and not self.base.is_posterior and self.fdd.n > 0 and hasattr(eng, "factor_dense_batch")
                and getattr(eng, "_deferred", None) is not None and getattr(eng, "cholesky_retry_factor", 1.0) <= 1.0
                and isinstance(self.y, torch.Tensor) and self.y.is_cuda)

    def factor(self):
        """Cholesky of cov(f(X)) + D + eps I under the observed process and L^-1 (y - mean(X))."""
        if self._fac is None:
            self._fac = self.fdd._factor(self.y)
        return self._fac

    def merged_with(self, other):
        """Observations of the prior equivalent to conditioning on `self` and then on `other` (both dense)."""
        prior = self.prior_gp()
        a, b = self.fdd, other.fdd
        na = a.noise if a.noise is not None else torch.zeros(a.n, dtype=torch.float64, device=a.x.device)
        nb = b.noise if b.noise is not None else torch.zeros(b.n, dtype=torch.float64, device=b.x.device)
        x = torch.cat([a.x, b.x], dim=0)
        return Obs(FDD(prior, x, torch.cat([na, nb])), torch.cat([self.y, other.y], dim=0))

    # ---- the posterior's corrections (any base process) -----------------------------------------------
    def _V(self, p):
        """V = k_base(p, X) L^-T  (n_p x n)."""
        V = self.base._cross(p, self.fdd.pts())
        self.eng.trsm_rlt_(self.factor().L, V)
        return V

    def mean_at(self, p):
        base_mean = self.base._mean_at(p)
        if self.fdd.n == 0:
            return base_mean
        Ks = self.base._cross(p, self.fdd.pts())
        corr = self.eng.gemm(Ks, self.factor().alpha(), tb=True)
        return corr if not self.base.is_posterior else base_mean + corr

    def cross(self, pa, pb):
        K = self.base._cross(pa, pb)
        if self.fdd.n:
            self.eng.gemm(self._V(pa), self._V(pb), tb=True, alpha=-1.0, beta=1.0, out=K)
        return K

    def diag(self, p):
        d = self.base._diag(p)
        return d - self.eng.rownorm2(self._V(p)) if self.fdd.n else d

    def moments_into(self, p, block, diag_add, jitter):
        eng = self.eng
        mean = self.base._moments_into(p, block, diag_add, jitter)
        if self.fdd.n == 0:
            return mean
        V = self._V(p)
        eng.gemm(V, V, tb=True, alpha=-1.0, beta=1.0, out=block, c_lower=True)
        corr = eng.gemm(V, self.factor().zrow, tb=True)
        return corr if not self.base.is_posterior else mean + corr

    # ---- batched routines for dense observations of the prior (`fast_dense`) ----------------------------
    def posterior_mean_batch(self, xs):
        """Posterior means at several input sets with one stacked cross-Gram product."""
        eng, fac = self.eng, self.factor()
        ck, z = self.fdd.features()
        sizes = [int(x_s.shape[0]) for x_s in xs]
        if self.fdd.n == 0:
            return [torch.zeros(k, 1, dtype=torch.float64, device=z.device) for k in sizes]
        B = eng.new_matrix(sum(sizes), self.fdd.n)
        stacked = xs.matrix if isinstance(xs, Stacked) else torch.cat([_as_matrix(eng, x_s) for x_s in xs], dim=0)
        eng.gram(ck, eng.features(ck, _as_matrix(eng, stacked)), z, out=B)   # one launch each
        means = eng.gemm(B, fac.alpha(), tb=True)
        return list(torch.split(means, sizes, dim=0))

    def _stacked_features(self, xs):
        """Features of several input sets of one size in ONE launch: (the stacked matrix, its per-set row views)."""
        eng = self.eng
        ck, _ = self.fdd.features()
        ns = int(xs[0].shape[0])
        stacked = xs.matrix if isinstance(xs, Stacked) else torch.cat([_as_matrix(eng, x_s) for x_s in xs], dim=0)
        z_all = eng.features(ck, _as_matrix(eng, stacked))
        return z_all, [z_all[k * ns : (k + 1) * ns] for k in range(len(xs))]

    def _stacked_V(self, z_all):
        """Rows [k ns, (k + 1) ns): V_k = K(x_k, X) L^-T for every feature set stacked in z_all - ONE cross-Gram launch, ONE
        triangular solve."""
        eng, fac = self.eng, self.factor()
        ck, z = self.fdd.features()
        B = eng.new_matrix(z_all.shape[0], self.fdd.n)
        eng.gram(ck, z_all, z, out=B)
        eng.trsm_rlt_(fac.L, B)
        return B

    def _chunk(self, ns):
        """Samples per chunk: the stacked right-hand side (S n* x n doubles) and the stacked per-sample covariances
        (S n* x n*) are each bounded to ~16 GB of the 288 GB HBM; at most 16384 samples at a time (the batched launches
        carry the sample index in a grid dimension, limit 65535)."""
        return max(1, min(16384, int(16e9 // max(1, ns * max(self.fdd.n, ns, 1) * 8))))

    def posterior_marginals_batch(self, xs):
        """(means, variances), each n* x S: column s holds the posterior mean / marginal variance (no noise) at xs[s]."""
        eng, fac = self.eng, self.factor()
        ck, z = self.fdd.features()
        n, S = self.fdd.n, len(xs)
        ns = int(xs[0].shape[0])
        mean = torch.zeros(ns, S, dtype=torch.float64, device=z.device)
        var = torch.empty(ns, S, dtype=torch.float64, device=z.device)
        chunk = self._chunk(ns)
        for s0 in range(0, S, chunk):
            s1 = min(S, s0 + chunk)
            z_all, zss = self._stacked_features(xs[s0:s1])
            kd = eng.gram_diag(ck, z_all).reshape(s1 - s0, ns).T
            if n > 0:
                B = self._stacked_V(z_all)
                mean[:, s0:s1] = eng.gemm(B, fac.zrow, tb=True).reshape(s1 - s0, ns).T
                kd = kd - eng.rownorm2(B).reshape(s1 - s0, ns).T
            var[:, s0:s1] = kd
        return mean, var

    def _sample_batch_linear_tail(self, xs, noise_vec, zr, out):
        """posterior_sample_batch when the columns that differ between the sets only enter the kernel through linear factors
        standing alone in their terms (kernels.linear_tail) - GPAR's DEFAULT output dependence (reference
        gpar/regression.py:141-146, 276-278: `linear=True, nonlinear=False`).  Then, with c_s the weighted features of those columns
        at set s and Z_c the same at the training inputs,
            K(x_s, X)   = K0(x, X) + c_s Z_c^T                       (K0: those features zeroed - the same for every s)
            V_s         = K(x_s, X) L^-T = V0 + c_s U^T,              U = L^-1 Z_c      (n x q, q = number of such features)
            mean_s      = V0 r + c_s (U^T r),                          r = L^-1 y
            cov_s       = [K0(x, x) + D - V0 V0^T] + c_s E_s^T + E_s c_s^T,    E_s = c_s (I - U^T U) / 2 - V0 U
        so the S stacked n* x n solves and the S rank-n downdates of the general routine become ONE of each plus a rank-2q update
        per set; what stays per set is the n* x n* factorisation (lock-step) and the draw.  Same law, same random numbers, same
        samples up to rounding.  Returns False (nothing done) when the structure or the sizes do not qualify."""
        from .kernels import linear_tail

        eng, fac = self.eng, self.factor()
        if os.environ.get("GPAR_LINEAR_TAIL", "1") == "0" or not isinstance(xs, Stacked) or xs.shared_cols <= 0:
            return False
        ck, z = self.fdd.features()
        n, S, ns = self.fdd.n, len(xs), xs.rows
        if n == 0 or S < 2 or not hasattr(eng, "potrf_batch_") or ns > eng.batch_rows() or getattr(eng._tls, "safe", False):
            return False
        tail = linear_tail(ck, xs.shared_cols)
        if tail is None:
            return False
        idx, weight = tail
        q = len(idx)
        dev = z.device
        sel = torch.as_tensor(idx, dtype=torch.long, device=dev)
        wts = torch.as_tensor(weight, dtype=torch.float64, device=dev)
        # shared part: the first set with the varying features zeroed
        z0 = eng.features(ck, _as_matrix(eng, xs[0])).clone()
        z0.index_fill_(1, sel, 0.0)
        V0 = eng.new_matrix(ns, n)
        eng.gram(ck, z0, z, out=V0)
        eng.trsm_rlt_(fac.L, V0)
        Ut = eng.new_matrix(q, n)                      # U^T = Z_c^T L^-T
        Ut.copy_((z.index_select(1, sel) * wts).T)
        eng.trsm_rlt_(fac.L, Ut)
        base = eng.new_matrix(ns, ns)
        eng.gram(ck, z0, lower=True, diag_add=noise_vec, diag_const=eng.epsilon, out=base)
        eng.gemm(V0, V0, tb=True, alpha=-1.0, beta=1.0, out=base, c_lower=True)
        H = eng.gemm(V0, Ut, tb=True)                  # V0 U: n* x q
        T = eng.gemm(Ut, Ut, tb=True)                  # U^T U: q x q
        mean0 = eng.gemm(V0, fac.zrow, tb=True)        # n* x 1
        beta = eng.gemm(Ut, fac.zrow, tb=True)         # q x 1
        half = 0.5 * (torch.eye(q, dtype=torch.float64, device=dev) - T)
        half = 0.5 * (half + half.T)                   # (symmetric to the last bit: c E^T + E c^T then is, too)
        chunk = max(1, min(16384, int(16e9 // max(1, ns * ns * 8))))
        for s0 in range(0, S, chunk):
            s1 = min(S, s0 + chunk)
            K = s1 - s0
            z_all = eng.features(ck, _as_matrix(eng, xs[s0:s1].matrix))
            C = eng.new_matrix(K * ns, q)
            C.copy_(z_all.index_select(1, sel) * wts)  # (K n*) x q
            E = eng.gemm(C, half) - H.repeat(K, 1)
            means = mean0.repeat(K, 1) + eng.gemm(C, beta)
            left = eng.new_matrix(K * ns, 2 * q)
            right = eng.new_matrix(K * ns, 2 * q)
            left[:, :q], left[:, q:] = C, E
            right[:, :q], right[:, q:] = E, C
            covs = eng.new_matrix(K * ns, ns)
            covs.unflatten(0, (K, ns)).copy_(base.unsqueeze(0))   # (a view: the rows of `covs` are padded)
            eng.gemm_batch_(left, right, covs, K, tb=True, alpha=1.0, beta=1.0, c_lower=True)
            _, info = eng.potrf_batch_(covs, K)
            eng.check_info(info)
            eng.trmv_lower_batch_(covs, K, zr[:, s0:s1], out[:, s0:s1], add=means)
        return True

    def posterior_sample_batch(self, xs, noise):
        eng, fac = self.eng, self.factor()
        ck, z = self.fdd.features()
        n, S = self.fdd.n, len(xs)
        ns = int(xs[0].shape[0])
        out = torch.empty(ns, S, dtype=torch.float64, device=z.device)
        if ns == 0:
            return out
        noise_vec = _noise_vector(eng, noise, ns)
        zr = eng.randn(ns, S)
        if self._sample_batch_linear_tail(xs, noise_vec, zr, out):
            return out
        chunk = self._chunk(ns)
        for s0 in range(0, S, chunk):
            s1 = min(S, s0 + chunk)
            z_all, zss = self._stacked_features(xs[s0:s1])
            B = means = None
            if n > 0:
                B = self._stacked_V(z_all)  # every V_s = K(x_s, X) L^-T in one solve
                means = eng.gemm(B, fac.zrow, tb=True)
            # the per-sample blocks (Gram, SYRK downdate, an n* x n* factorisation, one matvec) are independent and of one
            # shape.  Up to HipEngine.batch_rows() points they go through ONE batched downdate and ONE lock-step
            # factorisation (gpar_gemm_batch, gpar_potrf_batch: a small factorisation is a latency-bound chain, `count` of
            # them in step cost little more than one) ...
            K = s1 - s0
            if K > 1 and hasattr(eng, "potrf_batch_") and ns <= eng.batch_rows() and not getattr(eng._tls, "safe", False):
                covs = eng.new_matrix(K * ns, ns)
                eng.gram_batch_(ck, z_all, K, covs, lower=True, diag_add=noise_vec, diag_const=eng.epsilon)
                if n > 0:
                    eng.gemm_batch_(B, B, covs, K, tb=True, alpha=-1.0, beta=1.0, c_lower=True)
                _, info = eng.potrf_batch_(covs, K)
                eng.check_info(info)
                eng.trmv_lower_batch_(covs, K, zr[:, s0:s1], out[:, s0:s1], add=means)   # mean_s + chol(cov_s) z_s, all s
                continue
            # ... larger ones are dealt over a few streams and checked once at the end
            pipe = eng.pipeline(min(4, s1 - s0)) if s1 - s0 > 1 else None
            with eng.defer_checks(), joining(pipe):  # streams are joined BEFORE the deferred info words are read
                for k, zs in enumerate(zss):
                    with (pipe.stage(k, B, zr, out, zs, noise_vec, means) if pipe is not None else contextlib.nullcontext()):
                        cov = eng.new_matrix(ns, ns)
                        eng.gram(ck, zs, lower=True, diag_add=noise_vec, diag_const=eng.epsilon, out=cov)
                        mean = None
                        if n > 0:
                            V = B[k * ns : (k + 1) * ns]
                            eng.gemm(V, V, tb=True, alpha=-1.0, beta=1.0, out=cov, c_lower=True)
                            mean = means[k * ns : (k + 1) * ns]
                        _, info = eng.potrf_(cov)
                        eng.check_info(info)
                        draw = eng.trmv_lower(cov, zr[:, s0 + k : s0 + k + 1])
                        out[:, s0 + k : s0 + k + 1] = draw if mean is None else draw + mean
        return out


class _LazyV:
    """v = L_z^-T A^-1 c of an inducing-point observation, computed at first use on the stream of that use.  The producing
    stream and an event behind the solve are kept: a later consumer on ANOTHER stream (a posterior shared between worker streams)
    waits for that event instead of reading v unordered."""

    def __init__(self, eng, facA, Lz):
        self.eng, self.facA, self.Lz, self.value = eng, facA, Lz, None
        self._stream = self._ready = None

    def get(self):
        if self.value is None:
            v = self.facA.alpha().clone()  # (A^-1 c)^T, 1 x M
            self.eng.trsm_rln_(self.Lz, v)
            self.value = v
            if v.is_cuda:
                self._stream = torch.cuda.current_stream(v.device)
                self._ready = torch.cuda.Event()
                self._ready.record(self._stream)
        elif self._ready is not None:
            here = torch.cuda.current_stream(self.value.device)
            if here != self._stream:
                here.wait_event(self._ready)
                self.value.record_stream(here)
        return self.value


class PseudoObs:
    """Inducing-point observations (stheno `PseudoObs(f(x_ind), f(x, noise), y)`) of the process `fdd.p` (prior or
    posterior).  `method` selects the approximation, as stheno's `PseudoObsVFE` (the default, Titsias 2009) /
    `PseudoObsFITC` / `PseudoObsDTC` do:
        VFE   log N(y; m, Q + D) - 1/2 tr D^-1 (K - Q)
        DTC   log N(y; m, Q + D)
        FITC  log N(y; m, Q + D + diag(K - Q))        (the posterior then uses D + diag(K - Q) as well)
    with Q = K_xz K_zz^-1 K_zx."""

    fast_dense = False
    method = "vfe"

    def __init__(self, u, fdd, y):
        if isinstance(u, tuple):
            raise TypeError("pass a single FDD of inducing points")
        self.u = u
        self.fdd = fdd
        self.eng = fdd.eng
        self.base = fdd.p
        self.y = _as_matrix(self.eng, y)
        if u.p is not fdd.p and (u.p.kernel is not fdd.p.kernel or u.p.is_posterior or fdd.p.is_posterior):
            raise ValueError("inducing points and observations must belong to the same process")
        if fdd.noise is None:
            raise ValueError("inducing-point observations need observation noise")
        self._state = None

    def prior_gp(self):
        p = self.base
        while p.is_posterior:
            p = p._obs.base
        return p

    def rebased(self, gp):
        return type(self)(FDD(gp, self.u.x), FDD(gp, self.fdd.x, self.fdd.noise_arg), self.y)

    def _compute(self):
        if self._state is not None:
            return self._state
        eng, base = self.eng, self.base
        n, M = self.fdd.n, self.u.n
        px, pz = self.fdd.pts(), self.u.pts()
        d = self.fdd.noise
        # Which ORDER forms A - I = L_z^-1 K_zx D^-1 K_xz L_z^-T is decided on the device, from the pivot spread of L_z
        # (HipEngine.vfe_spread_limit): the n x M solve against L_z first (backward stable whatever cond(K_zz); M^2 n
        # flops), or the product first and the M x M result solved from both sides (2 M^3 flops; loses ~cond(K_zz) digits).
        # The gradient pass and FITC need the solved cross-Gram itself: they always take the first.
        limit = 0.0
        if self.method != "fitc" and not getattr(self, "_force_solve", False) and hasattr(eng, "vfe_spread_limit"):
            limit = eng.vfe_spread_limit(n, M)
        # L_z = chol(K_zz + eps I): a latency-bound chain of panel kernels on an M x M matrix (0.3 ms at M = 1024) that nothing
        # below needs before the triangular solve - on a side stream it runs beside the n x M cross-Gram build
        side = eng.side_stream() if hasattr(eng, "side_stream") and not base.is_posterior and n * M >= (1 << 22) else None
        Lz = eng.new_matrix(M, M)   # (allocated on the caller's stream: its lifetime follows that stream, whatever happens on the side)
        ill = None
        if side is not None:
            # Lz belongs to the caller's stream's pool but is written on the side stream: tell the allocator, so that an exception
            # between here and the join (out of memory in the n x M cross-Gram below, say) cannot hand the block to a new tensor
            # while the side stream's factorisation is still writing it
            Lz.record_stream(side)
        try:
            with (_on_side(side) if side is not None else contextlib.nullcontext()):
                mean_z = base._moments_into(pz, Lz, None, eng.epsilon)
                del mean_z  # the bound only involves the mean at the observed inputs
                _, info = eng.potrf_(Lz)
                if limit > 0.0:
                    _, ill = eng.chol_spread(Lz, limit)   # device word: 1 = K_zz too ill-conditioned for the product-first order
        except BaseException:
            if side is not None:   # whatever went wrong, the caller's stream is ordered after the side stream again
                torch.cuda.current_stream(side.device).wait_stream(side)
            raise
        pending_join = side is not None

        def joined():   # the factor is needed from here on (and its info word may only be read once it has been written)
            nonlocal pending_join
            if pending_join:
                _join_side(side, info, ill)
                pending_join = False
            eng.check_info(info)

        kdiag = base._diag(px)
        if self.method == "fitc":
            # the effective noise needs q_aa = |B_:a|^2 before anything can be scaled by it: one more pass over n x M
            Bs = base._cross(px, pz)
            joined()
            eng.trsm_rlt_(Lz, Bs)
            excess = kdiag - eng.rownorm2(Bs)  # k_aa - q_aa >= 0 up to rounding
            d = d + torch.clamp(excess, min=0.0)
            rs = torch.rsqrt(d)
            Bs.mul_(rs[:, None])
        else:
            rs = torch.rsqrt(d)
            # Bs = D^-1/2 K_xz L_z^-T (n x M): the row scaling rides along in the Gram kernel
            Bs = base._cross(px, pz, row_scale=rs)
            joined()
            if ill is None:
                eng.trsm_rlt_(Lz, Bs)
            else:
                eng.trsm_rlt_(Lz, Bs, when=(ill, True))   # (product-first order: Bs stays D^-1/2 K_xz)
        resid = self.y if not base.is_posterior else self.y - base._mean_at(px)
        ys = resid.reshape(-1) * rs
        # A - I = Bs^T Bs: ONE product over the n data points (K = n is cut into slices so that the whole chip works);
        # c = Bs^T ys is a matrix-vector pass; the trace term needs no pass of its own:
        #   sum_a (k_aa - |B_:a|^2) / d_a = sum_a k_aa / d_a - tr(A - I)
        # (the matrix-vector pass is bound by memory, the product by the matrix cores: side by side on two streams)
        side2 = eng.side_stream() if hasattr(eng, "side_stream") and n * M >= (1 << 22) else None
        if side2 is not None:
            with _on_side(side2):
                c = eng.gemv_t(Bs, ys).reshape(1, M)
            G = eng.gemm(Bs, Bs, ta=True, c_lower=True)
            _join_side(side2, c)
        else:
            G = eng.gemm(Bs, Bs, ta=True, c_lower=True)
            c = eng.gemv_t(Bs, ys).reshape(1, M)
        if ill is not None:
            # product-first order (skipped on the device when K_zz is ill-conditioned): [S; c^T] <- [S; c^T] L_z^-T, then the
            # transposed M x M block once more - L_z^-1 S L_z^-T and L_z^-1 c.  S enters symmetrised, so that the transposition
            # is the identity when the solves are skipped.
            W = eng.new_matrix(M + 1, M)
            low = torch.tril(G)
            W[:M].copy_(low + torch.tril(low, -1).T)
            W[M:].copy_(c)
            eng.trsm_rlt_(Lz, W, when=(ill, False))
            c = W[M:].clone()
            G = eng.new_matrix(M, M)
            G.copy_(W[:M].T)
            eng.trsm_rlt_(Lz, G, when=(ill, False))
        if hasattr(eng, "vfe_factor") and getattr(eng, "cholesky_retry_factor", 1.0) <= 1.0 and os.environ.get("GPAR_VFE_FUSED_SCALARS", "1") != "0":
            # the scalar side - four sums, the assembly of [[I + G, .], [c^T, 0]], the bound from its pieces - in three launches
            # around the factorisation instead of ~25 tensor operations (0.15 ms per layer at C4)
            Abuf, logdetA, infoA, elbo = eng.vfe_factor(G, c, ys, kdiag, d, self.method == "vfe")
            eng.check_info(infoA)
            facA = _Factor.from_batch(eng, M, Abuf, logdetA)
        else:
            yDy = torch.sum(ys * ys)
            trace_term = torch.sum(kdiag / d) - torch.sum(torch.diagonal(G)) if self.method == "vfe" else 0.0

            def fill(block, scale):
                block.copy_(G)
                # (first attempt: A as it is - its diagonal is >= 1; a retry of lab's ladder adds the grown part of the jitter,
                # so that every rung factors a different matrix)
                block.diagonal().add_(1.0 + (scale - 1.0) * eng.epsilon)

            facA = _Factor(eng, M, fill, c)
            elbo = -0.5 * (trace_term + torch.sum(torch.log(d)) + n * _LOG_2PI + facA.logdet[0] + yDy - facA.quad)
        # v = L_z^-T A^-1 c, so that the mean correction at x* is K_*z v: two single-row backward solves (0.17 ms at M = 1024) that
        # only a posterior MEAN needs - the value of the bound does not (the last layer of a log marginal likelihood never asks)
        deferring = getattr(eng, "_deferred", None) is not None
        self._state = {"Lz": Lz, "La": facA.L, "v": _LazyV(eng, facA, Lz), "elbo": elbo.detach() if deferring else elbo.detach().cpu(),
                       "Bs": Bs, "G": G, "facA": facA, "kdiag": kdiag, "ys": ys, "d": d, "solved": ill is None,
                       "moves": (excess > 0).to(d.dtype) if self.method == "fitc" else None}
        return self._state

    def _value(self):
        return self._compute()["elbo"]

    def _solved_state(self):
        """The state with Bs = D^-1/2 K_xz L_z^-T in it (what the gradient pass differentiates): computed again, cross-Gram solve
        first, should a value-only evaluation have left the order to the device."""
        st = self._compute()
        if not st["solved"]:
            self._state, self._force_solve = None, True
            st = self._compute()
        return st

    def logpdf(self):
        """The VFE bound; differentiable with respect to kernel parameters and noise when they carry a graph."""
        if not self.base.is_posterior and torch.is_grad_enabled():
            params = kernel_parameters(self.base.kernel)
            noise = self.fdd.noise_arg if _needs_grad(self.fdd.noise_arg) else None
            X = self.fdd.x if self.fdd.x.requires_grad else None
            Z = self.u.x if self.u.x.requires_grad else None
            if params or noise is not None or X is not None or Z is not None:
                self._params = params
                self._noise_device = None if noise is None else noise.device
                self._force_solve = True   # (the gradient pass needs the solved cross-Gram)
                return _LogMarginal.apply(self, noise, X, Z, *[p[3] for p in params])
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
        The forward pass keeps Bs = D^-1/2 B^T, so B^T = D^1/2 Bs throughout.  The three weighted sums over kernel
        derivatives are one fused device pass each (`kernel_grads_vfe`)."""
        vfe, fitc = self.method == "vfe", self.method == "fitc"
        eng = self.eng
        st = self._solved_state()
        n, M = self.fdd.n, self.u.n
        d = st["d"]  # the observation noise; FITC: the effective noise d + k_aa - q_aa
        rs = torch.rsqrt(d)
        Bs, facA, Lz, G = st["Bs"], st["facA"], st["Lz"], st["G"]
        a = facA.alpha()  # (A^-1 B D^-1 y)^T, 1 x M
        alpha = (st["ys"] - eng.gemm(Bs, a, tb=True).reshape(-1)) * rs  # S^-1 y
        beta = eng.gemv_t(Bs, alpha / rs).reshape(1, M)  # (B alpha)^T
        Ainv = eng.chol_inverse(facA.L)  # lower triangle of A^-1
        Ainv_full = torch.tril(Ainv) + torch.tril(Ainv, -1).T
        # W_fu
        T = eng.new_matrix(n, M)
        if vfe:
            T.copy_(Bs)
        else:
            T.zero_()  # DTC: no trace term, W_fu = G P^T = [alpha beta^T - D^-1 B^T A^-1] L_z^-1
        eng.gemm(Bs, Ainv_full, alpha=-1.0, beta=1.0, out=T)  # Bs (I - A^-1)   [DTC: -Bs A^-1]
        T.mul_(rs[:, None])  # D^-1 B^T (I - A^-1)
        T.add_(alpha[:, None] * beta)
        # per-point terms: |L_A^-1 B_:a|^2 / d_a^2 = |Es_a|^2 / d_a with Es = Bs L_A^-T, and q_aa / d_a^2 = |Bs_a|^2 / d_a
        E = eng.new_matrix(n, M)
        E.copy_(Bs)
        eng.trsm_rlt_(facA.L, E)
        if vfe:
            noise_grad = 0.5 * (alpha * alpha - 1.0 / d + (eng.rownorm2(E) - eng.rownorm2(Bs)) / d + st["kdiag"] / (d * d))
            wdiag = -0.5 / d
        else:
            noise_grad = 0.5 * (alpha * alpha - 1.0 / d + eng.rownorm2(E) / d)
            wdiag = torch.zeros_like(d)
        del E
        S = eng.new_matrix(M, M)
        S.copy_(beta.reshape(M, 1) * beta + Ainv_full)
        if vfe:
            S.add_(torch.tril(G) + torch.tril(G, -1).T)  # B D^-1 B^T = A - I (the trace term's share)
        S.diagonal().sub_(1.0)
        if fitc:
            # FITC is the DTC objective at the effective noise e_a = d_a + k_aa - q_aa, q_aa = K_az K_zz^-1 K_za, which moves with
            # the kernel: with g_a = dF / de_a (the DTC noise gradient above), sum_a g_a de_a adds
            #   g_a dk_aa                      -> wdiag = g,
            #   -2 g_a [K_az K_zz^-1] dK_za    -> W_fu -= 2 diag(g) B^T L_z^-1          (B^T = D^1/2 Bs),
            #   +g_a [P dK_zz P^T]_aa          -> W_uu += P diag(g) P^T, i.e. S -= 2 B diag(g) B^T
            # (points whose k_aa - q_aa was clamped at zero do not move).
            chain = noise_grad * st["moves"]
            wdiag = chain
            T.sub_((2.0 * chain / rs)[:, None] * Bs)
            eng.gemm(Bs * (chain * d)[:, None], Bs, ta=True, alpha=-2.0, beta=1.0, out=S)
        eng.trsm_rln_(Lz, T)  # ... L_z^-1
        # W_uu
        eng.trsm_rln_(Lz, S)  # S L_z^-1
        St = eng.new_matrix(M, M)
        St.copy_(S.T)
        eng.trsm_rln_(Lz, St)  # L_z^-T S L_z^-1 (symmetric)
        Wuu = eng.new_matrix(M, M)
        Wuu.copy_(-0.25 * (St + St.T))
        ck = self.fdd.pts().ck
        grads = eng.kernel_grads_vfe(ck, self.fdd.x.detach(), self.u.x.detach(), T, Wuu, wdiag)
        self._weights = (T, Wuu, wdiag)
        return noise_grad, grads

    def input_gradients(self, want_x, want_z):
        """dF / dX and dF / dZ from the same three weight arrays as the parameter gradients:
        F' = sum W_fu dK(X, Z) + sum W_uu dK(Z, Z) + sum wdiag dk(x, x)."""
        eng = self.eng
        T, Wuu, wdiag = self._weights
        ck = self.fdd.pts().ck
        X, Z = self.fdd.x.detach(), self.u.x.detach()
        gX = gZ = None
        if want_x:
            gX = eng.kernel_input_grads(ck, X, Z, T) + eng.kernel_diag_input_grads(ck, X, wdiag)
        if want_z:
            Tt = eng.new_matrix(Z.shape[0], X.shape[0])
            Tt.copy_(T.T)
            gZ = eng.kernel_input_grads(ck, Z, X, Tt) + eng.kernel_input_grads(ck, Z, None, Wuu, sym=True)
        return gX, gZ

    def mean_gradients(self, xs, g, want_xs=True, want_x=True, want_z=True):
        """Gradients of  g^T mean(xs),  mean(xs) = K(xs, Z) v,  v = Sigma^-1 K_zx D^-1 y,  Sigma = K_zz + K_zx D^-1 K_xz.
        With h = Sigma^-1 K(Z, xs) g, p = K_xz v (the mean at X) and q = K_xz h:
            d = sum (g v^T) dK(xs, Z) + sum W_fu dK(X, Z) + sum W_uu dK(Z, Z) - sum_a q_a (y_a - p_a) / d_a^2 dd_a,
            W_fu = ((y - p) / d) h^T - (q / d) v^T,     W_uu = -1/2 (h v^T + v h^T).
        FITC: d is the effective noise e_a = d_a + k_aa - q_aa; the term sum_a n_a de_a (n_a = the noise gradient above) adds
        n_a dk_aa - 2 n_a [K_az K_zz^-1] dK_za + n_a [P dK_zz P^T]_aa to the three weighted sums, as in `gradients`."""
        eng, st = self.eng, self._compute()
        ck = self.fdd.pts().ck
        X, Z = self.fdd.x.detach(), self.u.x.detach()
        n, M = self.fdd.n, self.u.n
        d = st["d"]  # FITC: the effective noise (its own dependence on the kernel is chained in below)
        v = st["v"].get().reshape(-1)
        pxs = self.base._pts(xs)
        Ksz = eng.gram(ck, pxs.z, self.u.pts().z)  # n* x M
        Kxz = eng.gram(ck, self.fdd.pts().z, self.u.pts().z)  # n x M
        h = eng.gemv_t(Ksz, g).reshape(1, M).clone()
        eng.trsm_rlt_(st["Lz"], h)
        eng.trsm_rlt_(st["La"], h)
        eng.trsm_rln_(st["La"], h)
        eng.trsm_rln_(st["Lz"], h)
        h = h.reshape(-1)
        p_ = eng.gemm(Kxz, v.reshape(1, M), tb=True).reshape(-1)
        q_ = eng.gemm(Kxz, h.reshape(1, M), tb=True).reshape(-1)
        resid = self.y.reshape(-1).detach() - p_
        W_sz = eng.new_matrix(xs.shape[0], M)
        W_sz.copy_(g[:, None] * v[None, :])
        W_fu = eng.new_matrix(n, M)
        W_fu.copy_((resid / d)[:, None] * h[None, :] - (q_ / d)[:, None] * v[None, :])
        W_uu = eng.new_matrix(M, M)
        W_uu.copy_(-0.5 * (h[:, None] * v[None, :] + v[:, None] * h[None, :]))
        noise_grad = -q_ * resid / (d * d)
        chain = None
        if self.method == "fitc":
            chain = noise_grad * st["moves"]
            Pt = eng.new_matrix(n, M)
            Pt.copy_(st["Bs"] * torch.sqrt(d)[:, None])
            eng.trsm_rln_(st["Lz"], Pt)  # K_xz K_zz^-1
            W_fu.sub_(2.0 * chain[:, None] * Pt)
            eng.gemm(Pt * chain[:, None], Pt, ta=True, alpha=1.0, beta=1.0, out=W_uu)
            del Pt
        params = _add_grads(eng.kernel_grads_weighted(ck, xs, Z, W_sz), eng.kernel_grads_weighted(ck, X, Z, W_fu))
        params = _add_grads(params, eng.kernel_grads_weighted(ck, Z, None, W_uu, sym=True))
        if chain is not None:
            params = _add_grads(params, eng.kernel_grads_diag(ck, X, chain))
        out = {"params": params, "noise": noise_grad}
        if want_xs:
            out["xs"] = eng.kernel_input_grads(ck, xs, Z, W_sz)
        if want_x:
            out["x"] = eng.kernel_input_grads(ck, X, Z, W_fu)
            if chain is not None:
                out["x"] = out["x"] + eng.kernel_diag_input_grads(ck, X, chain)
        if want_z:
            A_ = eng.new_matrix(M, xs.shape[0])
            A_.copy_(W_sz.T)
            B_ = eng.new_matrix(M, n)
            B_.copy_(W_fu.T)
            out["z"] = (eng.kernel_input_grads(ck, Z, xs, A_) + eng.kernel_input_grads(ck, Z, X, B_)
                        + eng.kernel_input_grads(ck, Z, None, W_uu, sym=True))
        return out

    # ---- the posterior's corrections (any base process) -----------------------------------------------
    def _P(self, p):
        """(k_base(p, Z), P = k_base(p, Z) L_z^-T) as two n_p x M matrices."""
        Kz = self.base._cross(p, self.u.pts())
        P = Kz.clone()
        self.eng.trsm_rlt_(self._compute()["Lz"], P)
        return Kz, P

    def mean_at(self, p):
        st = self._compute()
        Kz = self.base._cross(p, self.u.pts())
        corr = self.eng.gemm(Kz, st["v"].get(), tb=True)
        return corr if not self.base.is_posterior else self.base._mean_at(p) + corr

    # kept for callers that use the stheno-like name
    def posterior_mean(self, x):
        return self.mean_at(self.base._pts(x))

    def cross(self, pa, pb):
        eng, st = self.eng, self._compute()
        K = self.base._cross(pa, pb)
        _, Pa = self._P(pa)
        _, Pb = self._P(pb)
        eng.gemm(Pa, Pb, tb=True, alpha=-1.0, beta=1.0, out=K)
        eng.trsm_rlt_(st["La"], Pa)
        eng.trsm_rlt_(st["La"], Pb)
        eng.gemm(Pa, Pb, tb=True, alpha=1.0, beta=1.0, out=K)
        return K

    def diag(self, p):
        eng, st = self.eng, self._compute()
        _, P = self._P(p)
        out = self.base._diag(p) - eng.rownorm2(P)
        eng.trsm_rlt_(st["La"], P)
        return out + eng.rownorm2(P)

    # ---- batched sampling (ancestral sampling hands every layer S input sets of one size: GPAR.sample_many) ----------------
    @property
    def fast_sparse(self):
        """The batched routines below are written for inducing-point observations of a PRIOR process (what GPAR builds)."""
        return not self.base.is_posterior

    def _stacked_P(self, xs):
        """For the input sets stacked by rows: (features, K(x_s, Z), P_s = K(x_s, Z) L_z^-T), ONE launch each."""
        eng, st = self.eng, self._compute()
        ck, zu = self.u.features()
        stacked = xs.matrix if isinstance(xs, Stacked) else torch.cat([_as_matrix(eng, x_s) for x_s in xs], dim=0)
        z_all = eng.features(ck, _as_matrix(eng, stacked))
        Kz = eng.new_matrix(z_all.shape[0], self.u.n)
        eng.gram(ck, z_all, zu, out=Kz)
        P = Kz.clone()
        eng.trsm_rlt_(st["Lz"], P)
        return ck, z_all, Kz, P

    def _sparse_chunk(self, ns):
        return max(1, min(16384, int(16e9 // max(1, ns * max(self.u.n, ns, 1) * 8))))

    def posterior_mean_batch(self, xs):
        """Posterior means K(x_s, Z) v at several input sets with one stacked cross-Gram product."""
        eng, st = self.eng, self._compute()
        ck, zu = self.u.features()
        sizes = [int(x_s.shape[0]) for x_s in xs]
        stacked = xs.matrix if isinstance(xs, Stacked) else torch.cat([_as_matrix(eng, x_s) for x_s in xs], dim=0)
        Kz = eng.new_matrix(sum(sizes), self.u.n)
        eng.gram(ck, eng.features(ck, _as_matrix(eng, stacked)), zu, out=Kz)
        return list(torch.split(eng.gemm(Kz, st["v"].get(), tb=True), sizes, dim=0))

    def posterior_marginals_batch(self, xs):
        """(means, variances), each n* x S (no noise):  k_aa - |P_a|^2 + |P_a L_A^-T|^2  per point."""
        eng, st = self.eng, self._compute()
        S, ns = len(xs), int(xs[0].shape[0])
        dev = self.y.device
        mean = torch.zeros(ns, S, dtype=torch.float64, device=dev)
        var = torch.empty(ns, S, dtype=torch.float64, device=dev)
        chunk = self._sparse_chunk(ns)
        for s0 in range(0, S, chunk):
            s1 = min(S, s0 + chunk)
            ck, z_all, Kz, P = self._stacked_P(xs[s0:s1])
            mean[:, s0:s1] = eng.gemm(Kz, st["v"].get(), tb=True).reshape(s1 - s0, ns).T
            kd = eng.gram_diag(ck, z_all) - eng.rownorm2(P)
            eng.trsm_rlt_(st["La"], P)
            var[:, s0:s1] = (kd + eng.rownorm2(P)).reshape(s1 - s0, ns).T
        return mean, var

    def posterior_sample_batch(self, xs, noise):
        """One joint draw at each input set: mean_s + chol(K_ss - P_s P_s^T + R_s R_s^T + noise + eps I) z_s with P_s = K(x_s, Z) L_z^-T,
        R_s = P_s L_A^-T - the stacked cross-Gram, both triangular solves, the per-sample Gram builds, the two rank-M corrections
        and the n* x n* factorisations each ONE (batched / lock-step) launch sequence for all samples, as the dense posterior's
        sampler does it.  (Until round 3 every sample went through `f(x_s).sample()` on its own: `predict` with 50 samples at
        n* = 1000, M = 300 took 85 ms - four times the DENSE posterior's 21 ms.)"""
        eng, st = self.eng, self._compute()
        S, ns = len(xs), int(xs[0].shape[0])
        out = torch.empty(ns, S, dtype=torch.float64, device=self.y.device)
        if ns == 0:
            return out
        noise_vec = _noise_vector(eng, noise, ns)
        zr = eng.randn(ns, S)
        chunk = self._sparse_chunk(ns)
        for s0 in range(0, S, chunk):
            s1 = min(S, s0 + chunk)
            K = s1 - s0
            ck, z_all, Kz, P = self._stacked_P(xs[s0:s1])
            means = eng.gemm(Kz, st["v"].get(), tb=True)
            batched = K > 1 and hasattr(eng, "potrf_batch_") and ns <= eng.batch_rows() and not getattr(eng._tls, "safe", False)
            if batched:
                covs = eng.new_matrix(K * ns, ns)
                eng.gram_batch_(ck, z_all, K, covs, lower=True, diag_add=noise_vec, diag_const=eng.epsilon)
                eng.gemm_batch_(P, P, covs, K, tb=True, alpha=-1.0, beta=1.0, c_lower=True)
                eng.trsm_rlt_(st["La"], P)
                eng.gemm_batch_(P, P, covs, K, tb=True, alpha=1.0, beta=1.0, c_lower=True)
                _, info = eng.potrf_batch_(covs, K)
                eng.check_info(info)
                eng.trmv_lower_batch_(covs, K, zr[:, s0:s1], out[:, s0:s1], add=means)
                continue
            R = P.clone()
            eng.trsm_rlt_(st["La"], R)
            for k in range(K):
                rows = slice(k * ns, (k + 1) * ns)
                cov = eng.new_matrix(ns, ns)
                eng.gram(ck, z_all[rows], lower=True, diag_add=noise_vec, diag_const=eng.epsilon, out=cov)
                eng.gemm(P[rows], P[rows], tb=True, alpha=-1.0, beta=1.0, out=cov, c_lower=True)
                eng.gemm(R[rows], R[rows], tb=True, alpha=1.0, beta=1.0, out=cov, c_lower=True)
                _, info = eng.potrf_(cov)
                eng.check_info(info)
                out[:, s0 + k : s0 + k + 1] = eng.trmv_lower(cov, zr[:, s0 + k : s0 + k + 1]) + means[rows]
        return out

    def moments_into(self, p, block, diag_add, jitter):
        eng, st = self.eng, self._compute()
        mean = self.base._moments_into(p, block, diag_add, jitter)
        Kz, P = self._P(p)
        corr = eng.gemm(Kz, st["v"].get(), tb=True)
        eng.gemm(P, P, tb=True, alpha=-1.0, beta=1.0, out=block, c_lower=True)
        eng.trsm_rlt_(st["La"], P)  # Q = P L_A^-T
        eng.gemm(P, P, tb=True, alpha=1.0, beta=1.0, out=block, c_lower=True)
        return corr if not self.base.is_posterior else mean + corr


class PseudoObsFITC(PseudoObs):
    method = "fitc"


class PseudoObsDTC(PseudoObs):
    method = "dtc"


PseudoObsVFE = PseudoObs
SparseObs = PseudoObs
