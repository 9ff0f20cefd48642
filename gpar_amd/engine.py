"""The linear-algebra seam of gpar_amd: the set of device primitives the GP layer objects are written against.

In the reference this seam is `lab` + `matrix` + `mlkernels` dispatching to torch-CPU/LAPACK;
here the ONLY implementation shipped in the package is `HipEngine`, which forwards every primitive to
libgpar_hip.so (hand-written gfx950 kernels) on torch-allocated HBM buffers.  There is no CPU implementation in
this package and no fallback: constructing the default engine without the shared library or without a GPU
raises.  (The test-suite's CPU oracle implements the same small interface in numpy under `oracle/` so that the
host orchestration can be exercised without a GPU and the HIP results can be checked against it; it is never
imported from here.)

Primitive set (all tensors float64, row-major, lower triangles authoritative):
    tensor(x)                         host/array-like -> device tensor
    compile(kernel, width)            kernel algebra -> device kernel spec
    features(ck, x)                   z = stretch(periodic(select(x)))
    gram(ck, z1, z2=None, ...)        fused Gram (+ noise diagonal + jitter), optionally lower-only
    gram_diag(ck, z)                  k(x_i, x_i)
    new_matrix(r, c, zero=False)      workspace with aligned, padded rows
    potrf_(A, nf=None)                (partial) Cholesky -> (logdet, info) device scalars
    trsm_rlt_(L, B) / trsm_rln_(L, B) B L^-T / B L^-1
    gemm(A, B, ta, tb, alpha, beta, out, c_lower, a_lower)
    gemv_t(A, v) / rownorm2(A)        A^T v for a tall A / squared row norms (HBM-bound passes)
    randn(rows, cols)                 counter-based standard normals
    sample_stats(samples, qlo, qhi)   Monte-Carlo mean / percentiles over the sample axis
"""
import contextlib
import os
import threading

import torch

from . import _lib, hip
from .kernels import compile_kernel

__all__ = ["HipEngine", "get_engine", "set_engine", "NotPositiveDefiniteError", "HandOffTimeoutError", "joining"]


class NotPositiveDefiniteError(ArithmeticError):
    """Cholesky hit a non-positive pivot (LAPACK info > 0)."""

    def __init__(self, info):
        super().__init__(f"matrix is not positive definite: pivot {info} is not positive")
        self.info = info


class HandOffTimeoutError(ArithmeticError):
    """A workgroup of the persistent panel kernel gave up waiting for a tile from another one (device-side info < 0): the
    factorisation's results are invalid.  Waits only ever target workgroups dispatched earlier, so this needs an
    external cause (a hung or pre-empted GPU); callers retry once on the unfused path (`HipEngine.safe_mode`), and the
    optimiser treats it as a failed evaluation."""

    def __init__(self, code):
        super().__init__(f"gpar_potrf: device-side hand-off timed out (code {code})")
        self.code = code


_HW_QUEUES = None
# defaults of the library's thresholds (csrc/gram_jit.h, csrc/grad_jit.h): entries per launch from which kernels compiled at run time
# for the layer's structure are used, and the widest structure that has a generated Gram kernel
GRAM_JIT_MIN_ENTRIES = 1 << 26
GRAD_JIT_MIN_ENTRIES = 1 << 24
GRAM_JIT_MAX_DZ = 16        # up to here the strip kernel (dim loops unrolled); wider, up to GRAM_JIT_WIDE_MAX_DZ, the 4 x 4 micro-tile form
GRAM_JIT_WIDE_MAX_DZ = 48


def _hardware_queues():
    """Hardware queues the HIP runtime of this process maps streams onto (GPU_MAX_HW_QUEUES; the runtime's default is 4 and it
    reads the variable once, when it initialises).  Independent layers run on up to four streams beside the caller's and each
    factorisation may add a look-ahead side stream; with four queues the fourth layer stream silently runs behind the first
    (C2: 5.4 ms instead of 4.85).  So: a value the user exported always wins; if there is none and this process has not used
    the GPU through torch yet, 8 is exported here (engine creation, not package import) and assumed to take effect; if the
    GPU is already in use, the runtime's default is assumed and the layer pipeline stays at three streams."""
    global _HW_QUEUES
    if _HW_QUEUES is None:
        env = os.environ.get("GPU_MAX_HW_QUEUES")
        if env is not None:
            _HW_QUEUES = int(env)
        elif not torch.cuda.is_initialized():
            os.environ["GPU_MAX_HW_QUEUES"] = "8"
            _HW_QUEUES = 8
        else:
            _HW_QUEUES = 4
    return _HW_QUEUES


class HipEngine:
    name = "hip"

    def __init__(self, device=None, seed=0, epsilon=1e-12):
        _lib.load()  # raises HipLibraryError if libgpar_hip.so is missing: no fallback
        self.hw_queues = _hardware_queues()  # before the first GPU call below
        if not torch.cuda.is_available():
            raise RuntimeError(
                "gpar_amd needs an AMD GPU (gfx950): torch.cuda.is_available() is False and there is no CPU fallback"
            )
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.epsilon = float(epsilon)  # lab's B.epsilon: diagonal jitter added before every Cholesky
        # lab's B.cholesky_retry_factor: a failed Cholesky is retried with the jitter multiplied by 10 while the factor stays
        # <= this value (default 1: no retry, as in lab).  Retrying needs the verdict of each attempt, i.e. a host sync per
        # factorisation: layers are then neither pipelined nor checked in one go.
        self.cholesky_retry_factor = 1.0
        self._seed = int(seed)
        self._calls = 0
        self._tls = threading.local()  # per host thread: pending device-side info words of an open defer_checks() block
        self._prepared = set()         # (kernel structure, kind) whose run-time compiled device kernel has been requested

    # ---- memory ----------------------------------------------------------------------------------
    def tensor(self, x):
        if isinstance(x, torch.Tensor):
            if x.requires_grad and torch.is_grad_enabled():
                # an input that is a function of hyper-parameters (a posterior mean fed forward under fit(fix=False)):
                # the graph is kept, the layer objects differentiate through it (gp._PosteriorMean, gp._LogMarginal)
                return x.to(device=self.device, dtype=torch.float64)
            return x.detach().to(device=self.device, dtype=torch.float64)
        return torch.as_tensor(x, dtype=torch.float64).to(self.device)

    def new_matrix(self, rows, cols, zero=False):
        return hip.alloc_matrix(rows, cols, self.device, zero=zero)

    def _mat(self, a):
        """Make `a` acceptable to the kernels (float64, on device, unit inner stride)."""
        if a.dim() == 2 and a.shape[1] > 1 and a.stride(1) != 1:
            a = a.contiguous()
        return a

    # ---- kernels ---------------------------------------------------------------------------------
    def compile(self, kernel, width):
        # A kernel object is an immutable expression tree, but its hyper-parameters may be torch tensors that the caller
        # updates IN PLACE (an optimiser step, `t.fill_()`): the compiled specification bakes their values, so it is cached
        # per (width, identity and version counter of every tensor parameter) - a stale specification is never returned.
        # (The layer constructors memoised by the variable store hand the same object to every evaluation of an epoch.)
        cache = kernel.__dict__.setdefault("_compiled", {})
        stamp = kernel.stamp()
        hit = cache.get(width)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        ck = compile_kernel(kernel, width)
        cache[width] = (stamp, ck)
        return ck

    def prepare(self, kernels, rows, training=False, sparse=False, inputs=False, cols=None):
        """Compile, NOW and concurrently, the run-time specialised device kernels that evaluating (and, with `training`,
        differentiating) layers with these `(kernel, width)` pairs on `rows` data points will ask for (csrc/jit.h): each structure
        costs 0.3-0.6 s of hiprtc time at first use, and left to the first evaluation the p layers' structures compile one after
        the other inside it, under the library's lock.  `gpar_jit_prepare` compiles outside the lock, one host thread per
        structure.  `sparse`: rectangular weight passes as well; `inputs`: the input-gradient passes (joint objective, trainable
        inducing inputs).  Does nothing for problems below the thresholds from which the generated kernels are used."""
        import ctypes
        from concurrent.futures import ThreadPoolExecutor

        lib = _lib.load()
        gram_min = int(os.environ.get("GPAR_GRAM_JIT_MIN_ENTRIES", str(GRAM_JIT_MIN_ENTRIES)))
        grad_min = int(os.environ.get("GPAR_GRAD_JIT_MIN_ENTRIES", str(GRAD_JIT_MIN_ENTRIES)))
        entries = int(rows) * int(cols if cols else rows)
        todo = {}
        for kernel, width in kernels:
            ck = self.compile(kernel, width)
            structure = (ck.dz, int(ck.kspec.nterms)) + tuple(
                (int(f.type), int(f.term), int(f.off), int(f.nd)) for f in ck.kspec.factor[: int(ck.kspec.nfactors)])
            zd = 20 if self._periodic(ck) else 0
            kinds = []
            if 0 <= gram_min <= entries and ck.dz <= GRAM_JIT_WIDE_MAX_DZ:
                kinds.append(0)
            if training and 0 <= grad_min <= entries:
                kinds.append(1 + zd)
                if sparse:
                    kinds.append(11 + zd)
                if inputs and 1 <= ck.dz <= 20:
                    kinds += [2, 12]
            for kind in kinds:
                if (structure, kind) not in self._prepared:
                    todo[(structure, kind)] = (kind, ck)
        if not todo:
            return 0
        stream = hip.stream_ptr(self.device)

        def compile_one(item):
            kind, ck = item
            return lib.gpar_jit_prepare(kind, ctypes.byref(ck.kspec), ck.dz, stream)

        with ThreadPoolExecutor(max_workers=min(8, len(todo))) as pool:
            list(pool.map(compile_one, todo.values()))
        self._prepared.update(todo)
        return len(todo)

    def features(self, ck, x):
        return hip.featurize(ck, self._mat(x))

    def gram(self, ck, z1, z2=None, lower=False, diag_add=None, diag_const=0.0, out=None, row_scale=None):
        return hip.gram(ck, z1, z2, out=out, lower=lower, diag_add=diag_add, diag_const=diag_const, row_scale=row_scale)

    def gram_diag(self, ck, z):
        return hip.gram_diag(ck, z)

    def potrf_(self, A, nf=None):
        # inside a layer pipeline of three or more streams the factorisations overlap one another: no look-ahead stream each
        safe = getattr(self._tls, "safe", False)
        return hip.potrf_(A, nf=nf, lookahead=not safe and getattr(self._tls, "pipe_depth", 0) < 3, fused=not safe)

    def potrf_batch_(self, A, batch):
        """`batch` square matrices stacked by rows in A, factored in lock-step (logdets, info words)."""
        return hip.potrf_batch_(A, batch, fused=not getattr(self._tls, "safe", False))

    def gram_batch_(self, ck, z_all, batch, out, lower=False, diag_add=None, diag_const=0.0):
        return hip.gram_batch_(ck, z_all, batch, out, lower=lower, diag_add=diag_add, diag_const=diag_const)

    def trmv_lower_batch_(self, Ls, batch, X, out, add=None):
        return hip.trmv_lower_batch_(Ls, batch, X, out, add=add)

    def gemm_batch_(self, A, B, out, batch, ta=False, tb=False, alpha=1.0, beta=0.0, c_lower=False):
        return hip.gemm_batch_(A, B, out, batch, ta=ta, tb=tb, alpha=alpha, beta=beta, c_lower=c_lower)

    def logpdf_dense(self, ck, x, y, noise_diag, jitter):
        """One dense layer's log marginal likelihood in one library call (value as a 0-d device tensor, info word)."""
        safe = getattr(self._tls, "safe", False)
        depth = getattr(self._tls, "pipe_depth", 0)
        value, _, info, _ = hip.logpdf_dense(ck, self._mat(x), y, noise_diag, jitter, lookahead=not safe and depth < 3, fused=not safe)
        return value, info

    def logpdf_dense_batch(self, items, jitter):
        """The same for several layers of equal size that do not feed one another, factored in lock-step (values, info words)."""
        safe = getattr(self._tls, "safe", False)
        return hip.logpdf_dense_batch([(ck, self._mat(x), y, nd) for ck, x, y, nd in items], jitter, fused=not safe)

    def logpdf_dense_grad(self, ck, x, y, noise_diag, jitter):
        """One dense layer's objective and the ingredients of its gradient in one library call: (value as a 0-d device tensor,
        info word, a function returning (1/2 diag(W) on the device, kernel-parameter gradients) once the host needs them, the
        (n + 1) x (n + 1) factor buffer with its log-determinant word - the factor a later posterior mean or conditioning on the
        same observations would otherwise compute a second time)."""
        safe = getattr(self._tls, "safe", False)
        depth = getattr(self._tls, "pipe_depth", 0)
        out, half_diag, info, A = hip.logpdf_dense_grad(ck, self._mat(x), y, noise_diag, jitter, self._periodic(ck),
                                                        lookahead=not safe and depth < 3, fused=not safe)

        def gradients():
            raw = out[2:].cpu().numpy()   # (the one device-to-host copy of the backward pass)
            return half_diag, self._grads_from_moments(ck, raw, 0.5)

        return out[0], info, gradients, (A, out[1:2])

    def logpdf_lockstep(self, layers, x, y, w, jitter):
        """The whole lock-step evaluation in one library call: (values, their sum in layer order, info words)."""
        safe = getattr(self._tls, "safe", False)
        return hip.logpdf_lockstep(layers, self._mat(x), self._mat(y), None if w is None else self._mat(w), jitter, fused=not safe)

    def factor_dense_batch(self, items, jitter):
        """The lock-step factorisations alone: (buffer of the `batch` augmented factors, logdets, info words)."""
        safe = getattr(self._tls, "safe", False)
        return hip.factor_dense_batch([(ck, self._mat(x), y, nd) for ck, x, y, nd in items], jitter, fused=not safe)

    def batch_rows(self):
        """Layers with at most this many rows are factored in lock-step rather than on separate streams (GPAR_LAYER_BATCH_ROWS;
        0 = never): below ~4600 rows a factorisation is a chain of latency-bound panel kernels (no look-ahead, no grouping), and
        several chains on separate streams fight for compute-unit slots.  Measured per evaluation, streams -> lock-step: four layers
        at n = 512 0.94 -> 0.61 ms, 2048 1.81 -> 1.34, 4096 (C2) 4.2 -> 3.3; eight at 2048 3.06 -> 1.86; sixteen at 8192 (C5) 65.0 -> 60.2;
        eight at 16384 (C3: a 17 GB batch) 195.5 -> 188.0.  Not measured above 16384 rows: the default stops at 20480."""
        return int(os.environ.get("GPAR_LAYER_BATCH_ROWS", "20480"))

    def batch_bytes(self):
        """Workspace budget of one lock-step batch (GPAR_LAYER_BATCH_BYTES, default 24 GiB): more layers than fit are factored in
        several batches."""
        return int(os.environ.get("GPAR_LAYER_BATCH_BYTES", str(24 << 30)))

    @contextlib.contextmanager
    def safe_mode(self):
        """Factorisations inside use the unfused panel path (separate leaf kernels, nothing waits inside a launch), without
        look-ahead and without layer pipelining: the retry path after a HandOffTimeoutError."""
        outer = getattr(self._tls, "safe", False)
        self._tls.safe = True
        try:
            yield
        finally:
            self._tls.safe = outer

    def trsm_rlt_(self, L, B, when=None):
        return hip.trsm_rlt_(L, B, when=when)

    def vfe_factor(self, G, c, ys, kdiag, d, with_trace):
        """chol of A = I + G with the row c appended and the inducing-point bound, the scalar side in two launches (hip.vfe_factor)."""
        safe = getattr(self._tls, "safe", False)
        return hip.vfe_factor(G, c, ys, kdiag, d, 1.0, with_trace, lookahead=not safe and getattr(self._tls, "pipe_depth", 0) < 3, fused=not safe)

    def chol_spread(self, L, limit):
        return hip.chol_spread(L, limit)

    def vfe_spread_limit(self, n, M):
        """Pivot spread of L_z = chol(K_zz) up to which the inducing-point bound forms K_zx D^-1 K_xz first and solves the M x M
        result against L_z from both sides, instead of solving the n x M cross-Gram - GPAR_VFE_SPREAD_MAX, DEFAULT 0 = never (the
        order stheno uses, backward stable whatever cond(K_zz)).  Opt-in because of what it costs in digits: the product-first
        order loses ~cond(K_zz), and the pivot spread only bounds cond(K_zz) from below (M = 512 uniform points in 8 dimensions:
        spread 53, cond 7.7e3).  Measured against 80-bit arithmetic (tools/exp_vfe_routes.py, profiles/r04_vfe_routes.txt):
        relative error of the bound 2e-16 .. 8e-13 for spread <= 3e3 at M = 96, 8e-13 at M = 512 / n = 8192; C4 (M = 1024,
        n = 65536) differs from the solve-first value by 2.2e-10 - where solve-first stays at 1e-15 .. 1e-14 throughout.  With a
        limit set (1e3 is what the measurements used) the order is decided on the device per evaluation (gpar_trsm_rlt_if: no
        host synchronisation, a failed or ill-conditioned factor takes the solve-first order, same bits as without the switch);
        C4 15.2 -> 12.2 ms.  Only offered where the n x M solve is large; gradient passes and FITC always solve first."""
        limit = float(os.environ.get("GPAR_VFE_SPREAD_MAX", "0"))
        if limit <= 0.0 or n * M < (1 << 24) or M < 128:
            return 0.0
        return limit

    def trsm_rln_(self, L, B):
        return hip.trsm_rln_(L, B)

    def gemm(self, A, B, ta=False, tb=False, alpha=1.0, beta=0.0, out=None, c_lower=False, a_lower=False):
        return hip.gemm(self._mat(A), self._mat(B), ta=ta, tb=tb, alpha=alpha, beta=beta, out=out, c_lower=c_lower, a_lower=a_lower)

    def chol_inverse(self, L):
        return hip.chol_inverse(L)

    def chol_inverse_x(self, L):
        """(lower triangle of (L L^T)^-1, X = L^-T in the upper triangle of the call's workspace)."""
        return hip.chol_inverse(L, with_x=True)

    def trmv_upper(self, U, x):
        """U x for an upper-triangular U and one vector (n entries), as a 1 x n row."""
        return hip.trmv_upper(self._mat(U), x).reshape(1, -1)

    def gemv_t(self, A, v):
        """A^T v for a tall matrix A and one weight per row (vector of A.shape[1] entries)."""
        return hip.gemv_t(self._mat(A), v)

    def rownorm2(self, A):
        """Squared Euclidean norm of every row of A (vector)."""
        return hip.rownorm2(self._mat(A))

    def pack_lower(self, A, out=None):
        """Lower triangle of A as n (n + 1) / 2 contiguous doubles (the exchange format of factors between ranks)."""
        return hip.pack_lower(A, out)

    def unpack_lower_(self, packed, A):
        return hip.unpack_lower_(packed, A)

    def trmv_lower(self, L, x):
        """L x for lower-triangular L and a single column x."""
        return hip.trmv_lower(self._mat(L), x)

    def kernel_grads(self, ck, x, W):
        """1/2 sum_ab W_ab dK_ab/dtheta for every parameter of the compiled kernel (W: lower triangle of a symmetric
        matrix).  One fused device pass produces per-term / per-factor / per-feature moment sums (csrc/gram.h); the
        chain rule from features to length scales, periods and alphas is applied here on the host."""
        x = self._mat(x)
        z = hip.featurize(ck, x)
        zd = hip.featurize_dfreq(ck, x) if self._periodic(ck) else None
        raw = hip.gram_grad(ck, z, zd, W).cpu().numpy()
        return self._grads_from_moments(ck, raw, 0.5)

    def kernel_diag_input_grads(self, ck, x, w):
        """d / d x of  sum_a w_a k(x_a, x_a)  (n x width).  EQ / RQ factors are 1 on the diagonal: only products of linear
        factors move with x."""
        x = self._mat(x).detach()
        out = torch.zeros(x.shape[0], ck.width, dtype=torch.float64, device=x.device)
        z = hip.featurize(ck, x)
        fs = ck.fspec
        by_term = {}
        for ti, fi, off, nd in ck.layout:
            if ck.kernel.terms[ti].factors[fi].type == "linear":
                by_term.setdefault(ti, []).append((off, nd))
        for ti, factors in by_term.items():
            coef = float(ck.kspec.coef[ti])
            values = [torch.sum(z[:, off : off + nd] ** 2, dim=1) for off, nd in factors]
            for k, (off, nd) in enumerate(factors):
                rest = w * coef
                for k2, v in enumerate(values):
                    if k2 != k:
                        rest = rest * v
                for q in range(off, off + nd):
                    out[:, int(fs.col[q])] += rest * 2.0 * z[:, q] * float(fs.inv_scale[q])  # linear factors are never periodic
        return out

    def kernel_grads_weighted(self, ck, x1, x2, W, sym=False):
        """sum_ab W_ab dK(x1_a, x2_b)/dtheta for every kernel parameter (full sums, no factor 1/2).  sym: x2 is x1 and W is
        symmetric, given by its lower triangle."""
        x1 = self._mat(x1)
        f1 = hip.featurize(ck, x1)
        periodic = self._periodic(ck)
        d1 = hip.featurize_dfreq(ck, x1) if periodic else None
        if sym:
            raw = hip.gram_grad_cross(ck, f1, d1, f1, d1, self._mat(W), hip.GRAD_SYM)
        else:
            x2 = self._mat(x2)
            f2 = hip.featurize(ck, x2)
            d2 = hip.featurize_dfreq(ck, x2) if periodic else None
            raw = hip.gram_grad_cross(ck, f1, d1, f2, d2, self._mat(W), hip.GRAD_RECT)
        return self._grads_from_moments(ck, raw.cpu().numpy(), 1.0)

    def kernel_grads_vfe(self, ck, x, z, W_fu, W_uu, wdiag):
        """sum_aj W_fu[a, j] dK(x_a, z_j) + sum_ij W_uu[i, j] dK(z_i, z_j) + sum_a wdiag[a] dk(x_a, x_a) for every kernel
        parameter (the gradient of the inducing-point bound, gp.PseudoObs.gradients): three fused device passes whose
        moment sums add, then the same host chain rule.  W_uu is symmetric (its lower triangle is read)."""
        x, zp = self._mat(x), self._mat(z)
        fx, fz = hip.featurize(ck, x), hip.featurize(ck, zp)
        periodic = self._periodic(ck)
        dx = hip.featurize_dfreq(ck, x) if periodic else None
        dz = hip.featurize_dfreq(ck, zp) if periodic else None
        raw = hip.gram_grad_cross(ck, fx, dx, fz, dz, self._mat(W_fu), hip.GRAD_RECT)
        raw = raw + hip.gram_grad_cross(ck, fz, dz, fz, dz, self._mat(W_uu), hip.GRAD_SYM)
        raw = raw + hip.gram_grad_cross(ck, fx, dx, fx, dx, wdiag.contiguous(), hip.GRAD_DIAG)
        return self._grads_from_moments(ck, raw.cpu().numpy(), 1.0)

    def kernel_grads_diag(self, ck, x, wdiag):
        """sum_a wdiag[a] dk(x_a, x_a) for every kernel parameter (one device pass over the n diagonal pairs)."""
        x = self._mat(x)
        fx = hip.featurize(ck, x)
        dx = hip.featurize_dfreq(ck, x) if self._periodic(ck) else None
        raw = hip.gram_grad_cross(ck, fx, dx, fx, dx, wdiag.contiguous(), hip.GRAD_DIAG)
        return self._grads_from_moments(ck, raw.cpu().numpy(), 1.0)

    def kernel_input_grads(self, ck, x1, x2, W, sym=False):
        """d / d x1 of  sum_ab W_ab k(x1_a, x2_b)  as an n1 x width matrix (x2 held fixed).  `sym`: x2 is x1, W is symmetric
        and given by its lower triangle, and BOTH arguments move: d / d x of sum_ab W_ab k(x_a, x_b) = 2 sum_b W_ab d_1 k.
        The device pass works in feature space (csrc/gram.h: gram_input_grad_kernel); the chain back to design-matrix
        columns - 1 / scale, and the derivative of the periodic embedding - is applied here."""
        x1 = self._mat(x1)
        z1 = hip.featurize(ck, x1)
        z2 = z1 if sym else hip.featurize(ck, self._mat(x2))
        gz = hip.gram_input_grad(ck, z1, z2, self._mat(W), hip.GRAD_SYM if sym else hip.GRAD_RECT)
        out = torch.zeros(x1.shape[0], ck.width, dtype=torch.float64, device=x1.device)
        fs = ck.fspec
        for q in range(ck.dz):
            c, inv, freq, embed = int(fs.col[q]), float(fs.inv_scale[q]), float(fs.freq[q]), int(fs.embed[q])
            if embed == _lib.EMBED_SIN:
                dz = inv * freq * torch.cos(freq * x1[:, c])
            elif embed == _lib.EMBED_COS:
                dz = -inv * freq * torch.sin(freq * x1[:, c])
            else:
                dz = inv
            out[:, c] += gz[:, q] * dz
        return 2.0 * out if sym else out

    @staticmethod
    def _periodic(ck):
        return any(f.periods is not None for t in ck.kernel.terms for f in t.factors)

    @staticmethod
    def _grads_from_moments(ck, raw, scale):
        """Host chain rule: moment sums (csrc/gram.h) -> d/d coefficient, length scales, periods, RQ alphas."""
        import numpy as np

        nT, nF, nD = _lib.GPAR_MAX_TERMS, _lib.GPAR_MAX_FACTORS, _lib.GPAR_MAX_DIMS
        C, Al = raw[:nT], raw[nT : nT + nF]
        A, P = raw[nT + nF : nT + nF + nD], raw[nT + nF + nD :]
        two = 2.0 * scale
        out = {"coef": [scale * C[t] for t in range(len(ck.kernel.terms))], "factors": [[] for _ in ck.kernel.terms]}
        for flat, (ti, fi, off, nd) in enumerate(ck.layout):
            f = ck.kernel.terms[ti].factors[fi]
            scales = f.scales_value()
            g = {"scales": -two * A[off : off + nd] / scales, "periods": None, "alpha": None}
            if f.type == "rq":
                g["alpha"] = scale * Al[flat]
            if f.periods is not None:
                periods = f.periods_value()
                ncol = len(f.cols)
                g["periods"] = -two * (P[off : off + ncol] + P[off + ncol : off + 2 * ncol]) * 2.0 * np.pi / periods**2
            out["factors"][ti].append(g)
        return out

    # ---- randomness ------------------------------------------------------------------------------
    def seed(self, seed):
        self._seed = int(seed)
        self._calls = 0

    def randn(self, rows, cols):
        out = hip.randn(self._seed, self._calls, rows, cols, self.device)
        self._calls += 1
        return out

    def sample_stats(self, samples, q_lo=None, q_hi=None):
        """(mean, lo-percentile, hi-percentile) over the leading (sample) axis, on the device."""
        return hip.sample_stats(samples.contiguous(), q_lo, q_hi)

    # ---- layer pipelining ------------------------------------------------------------------------
    def pipeline(self, depth=None, rows=None):
        """Streams for layers that do not depend on one another (complete data, no `replace`, no inducing points):
        the tail of a blocked factorisation is a latency-bound chain of small panels that leaves most of the chip
        idle, the front of the next one is throughput-bound - on alternating streams the two overlap.  `rows` (the
        problem size) picks the default depth: four streams below 9216 rows (with GPU_MAX_HW_QUEUES = 8, see the package
        __init__: C2 5.35 -> 4.85 ms, C5 75 -> 70.5 ms against three), two above; round-1 measurements, taken with the
        runtime's default of four hardware queues, when a fourth stream silently shared a queue: three streams paid while a factorisation is mostly latency-bound
        (measured per 8-layer evaluation with 2 / 3 / 4 streams: n = 2048 6.0 / 4.9 / 6.3 ms, n = 4096 12.5 / 10.3 / 12.9,
        n = 6144 22.6 / 22.2 / 25.8; at n = 8192 it depends on the kernel - 16 layers of the C3 kernel 77.4 / 81.1 / 86.6 ms,
        of C5's periodic + RQ kernel, whose Gram build is heavier, 82.1 / 79.5 - and the named config decides; C3 at 16384
        208 / 212).  Returns None
        when disabled (GPAR_LAYER_PIPELINE=0 or 1); any other value of the variable fixes the depth."""
        env = os.environ.get("GPAR_LAYER_PIPELINE")
        if (env is not None and int(env) < 2) or getattr(self._tls, "safe", False) or self.cholesky_retry_factor > 1:
            return None
        if depth is None:
            depth = int(env) if env is not None else ((4 if self.hw_queues >= 8 else 3) if rows is not None and rows < 9216 else 2)
        if depth < 2:
            return None
        return _LayerPipeline(self, _device_streams(self.device, depth))

    def side_stream(self):
        """One extra stream for a short piece of work that is independent of what the caller enqueues next (the factorisation of
        K_zz beside the cross-Gram build of the inducing-point path); None while the engine is in safe mode, inside a layer
        pipeline stage, or when disabled (GPAR_SIDE_STREAM=0)."""
        if getattr(self._tls, "safe", False) or getattr(self._tls, "pipe_depth", 0) or os.environ.get("GPAR_SIDE_STREAM", "1") == "0":
            return None
        # a stream of its own (not one of the layer / worker streams: from a worker thread whose current stream is that pool
        # entry the "side" work would silently serialise with the caller's)
        side = _SIDE.get(str(self.device))
        if side is None:
            side = _SIDE[str(self.device)] = torch.cuda.Stream(device=self.device)
        return None if side == torch.cuda.current_stream(self.device) else side

    def worker_streams(self, depth=None, rows=None):
        """The same streams, for callers that drive them from separate host threads (GPARRegressor.fit trains
        independent layers concurrently).  Empty when disabled (GPAR_FIT_THREADS=0/1).  Default: two threads; four where one
        layer's evaluation is a latency-bound chain long enough to matter and short enough to leave the chip idle - 2048 to 5120
        rows (fit(iters=20), four layers, 2 -> 4 threads: n = 2048 139 -> 120 ms, 3072 200 -> 135, 4096 274 -> 196, eight
        layers at 4096 590 -> 414; below, the threads contend for the interpreter: n = 400 89 -> 122; above, two evaluations
        fill the chip: n = 6144 625 -> 692, 8192 1063 -> 1149, C3 13.4 -> 13.3 s; profiles/r04_fit_threads.txt)."""
        if depth is None:
            env = os.environ.get("GPAR_FIT_THREADS")
            depth = int(env) if env is not None else (4 if rows is not None and 2048 <= int(rows) <= 5120 else 2)
        if depth < 2:
            return []
        return _device_streams(self.device, depth)

    # ---- status ----------------------------------------------------------------------------------
    def defer_checks(self):
        """Context manager: inside it `check_info` only records the device-side info words (no host sync), so a
        whole multi-layer evaluation is enqueued back to back; leaving the block synchronises once and raises if
        any factorisation failed."""
        return _Deferred(self)

    def check_info(self, info):
        """Synchronise on the device-side LAPACK-style info word and raise if a pivot failed (or record it while a
        defer_checks() block is open)."""
        pending = self._deferred
        if pending is not None:
            pending.append(info)
            return
        self._raise_for(info)

    @property
    def _deferred(self):
        return getattr(self._tls, "deferred", None)

    @_deferred.setter
    def _deferred(self, value):
        self._tls.deferred = value

    @staticmethod
    def _raise_for(info):
        """`info`: one word or several (a lock-step batch, the words gathered over an evaluation): the first failure is reported."""
        for code in info.reshape(-1).tolist():
            if code < 0:
                raise HandOffTimeoutError(int(code))
            if code != 0:
                raise NotPositiveDefiniteError(int(code))


_SIDE = {}     # device -> the one side stream of HipEngine.side_stream
_STREAMS = {}  # device -> extra streams, shared by every engine of the process (the library pairs each caller stream
               # with an internal side stream, so the set of caller streams is kept small and stable)


def _device_streams(device, depth):
    pool = _STREAMS.setdefault(str(device), [])
    while len(pool) < depth:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:depth]


@contextlib.contextmanager
def joining(pipe):
    """Whatever happens inside (a failed factorisation raises at the end of a deferred-check block), the stage streams of
    `pipe` (may be None) are joined back into the caller's stream before control returns."""
    try:
        yield pipe
    finally:
        if pipe is not None:
            pipe.join()


class _LayerPipeline:
    """`with pipe.stage(i, *inputs): ...` runs the block on stream i mod depth after everything enqueued on the
    caller's stream so far; `inputs` (tensors allocated on the caller's stream that the block reads) are kept alive
    until `join()`, which makes the caller's stream wait for every stage.  Tensors created inside a stage belong to
    that stage's stream (torch's caching allocator is stream-aware) and may be used by the caller after `join()`."""

    def __init__(self, engine, streams):
        self.engine = engine
        self.main = torch.cuda.current_stream(engine.device)
        self.streams = streams
        self.keep = []
        self.used = []

    @contextlib.contextmanager
    def stage(self, i, *inputs):
        s = self.streams[i % len(self.streams)]
        s.wait_stream(self.main)
        self.keep.extend(inputs)
        if s not in self.used:
            self.used.append(s)
        tls = self.engine._tls
        outer = getattr(tls, "pipe_depth", 0)
        tls.pipe_depth = len(self.streams)
        try:
            with torch.cuda.stream(s):
                yield
        finally:
            tls.pipe_depth = outer

    def keep_alive(self, *tensors):
        self.keep.extend(tensors)

    def join(self):
        for s in self.used:
            self.main.wait_stream(s)
        self.keep = []
        self.used = []


class _Deferred:
    def __init__(self, eng):
        self.eng = eng
        self.outer = None

    def __enter__(self):
        self.outer = self.eng._deferred
        if self.outer is None:
            self.eng._deferred = []
        return self

    def __exit__(self, exc_type, exc, tb):
        if self.outer is None:
            pending, self.eng._deferred = self.eng._deferred, None
            if exc_type is None and pending:
                # one device-to-host copy for the whole evaluation (each read of a word is a synchronising copy of its own)
                if len(pending) > 1 and all(isinstance(i, torch.Tensor) and i.is_cuda and i.dtype == pending[0].dtype for i in pending):
                    pending = [torch.cat([i.reshape(-1) for i in pending])]
                for info in pending:
                    self.eng._raise_for(info)
        return False

    @property
    def active(self):
        return True


_engine = None


def get_engine():
    """The process-wide engine; created on first use as a HipEngine on the current GPU (raises without one)."""
    global _engine
    if _engine is None:
        _engine = HipEngine()
    return _engine


def set_engine(engine):
    """Install an engine object (used by the test-suite to inject its CPU oracle, and to pick a device)."""
    global _engine
    previous, _engine = _engine, engine
    return previous
