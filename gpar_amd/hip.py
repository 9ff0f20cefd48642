"""Thin torch-tensor front end of the C ABI: tensors in, raw device pointers across the boundary.

PyTorch is used only as the device-memory allocator and stream provider; every numerical operation here is a
call into libgpar_hip.so on the tensor's `data_ptr()`.  All matrices are float64, row-major with unit inner
stride (`stride(1) == 1`); the leading dimension is `stride(0)`.
"""
import ctypes
import math
import os

import torch

from . import _lib

__all__ = [
    "stream_ptr",
    "alloc_matrix",
    "featurize",
    "gram",
    "gram_diag",
    "potrf_",
    "trsm_rlt_",
    "trsm_rln_",
    "gemm",
    "dot",
    "gemv_t",
    "rownorm2",
    "pack_lower",
    "unpack_lower_",
    "randn",
    "sample_stats",
    "trmv_lower",
    "gram_grad",
    "gram_grad_cross",
    "gram_input_grad",
    "chol_inverse",
]


def _check_mat(a, name):
    if a.dtype != torch.float64 or not a.is_cuda:
        raise TypeError(f"{name} must be a float64 tensor on the GPU (got {a.dtype} on {a.device})")
    if a.dim() == 2:
        if a.shape[1] > 1 and a.stride(1) != 1:
            raise ValueError(f"{name} must have unit inner stride")
    elif a.dim() != 1:
        raise ValueError(f"{name} must be a vector or a matrix")


def _ld(a):
    if a.dim() == 1:
        return 1
    # a single-row matrix may report any stride(0); make it safe for the kernels' index math
    return max(int(a.stride(0)), int(a.shape[1]), 1) if a.shape[0] > 1 else max(int(a.shape[1]), 1)


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def alloc_matrix(rows, cols, device, zero=False):
    """rows x cols view into a buffer whose leading dimension is padded to a multiple of 16 doubles (128-byte
    rows: every row start is aligned for the 16-byte vector paths and full-line stores)."""
    ld = max(16, (cols + 15) // 16 * 16)
    buf = (torch.zeros if zero else torch.empty)((max(rows, 1), ld), dtype=torch.float64, device=device)
    return buf[:rows, :cols]


def featurize(ck, x):
    """z = features(x) for a CompiledKernel `ck`; x: n x width."""
    _check_mat(x, "x")
    lib = _lib.load()
    n = x.shape[0]
    z = alloc_matrix(n, max(ck.dz, 1), x.device)
    if ck.dz == 0:
        z.zero_()
        return z
    _lib.check(
        lib.gpar_featurize(ctypes.byref(ck.fspec), x.data_ptr(), n, _ld(x), z.data_ptr(), _ld(z), stream_ptr(x.device)),
        "gpar_featurize",
    )
    return z


def logpdf_dense(ck, x, y, noise_diag, jitter, lookahead=True, fused=True):
    """log N(y; 0, k(x, x) + diag(noise_diag) + jitter I) in one library call (gpar_logpdf_dense): returns
    (value, logdet, info) as one-element device tensors and the (n + 1) x (n + 1) factor buffer."""
    lib = _lib.load()
    _check_mat(x, "x")
    n = x.shape[0]
    y = y.reshape(-1)
    if y.numel() != n or y.dtype != torch.float64 or not y.is_cuda:
        raise ValueError("y must hold one fp64 device value per row of x")
    nptr = None
    if noise_diag is not None:
        noise_diag = noise_diag.reshape(-1).contiguous()
        if noise_diag.numel() != n:
            raise ValueError("noise_diag must hold one value per row of x")
        nptr = noise_diag.data_ptr()
    z = alloc_matrix(n, max(ck.dz, 1), x.device)
    A = alloc_matrix(n + 1, n + 1, x.device)
    words = torch.empty(2, dtype=torch.float64, device=x.device)   # value, logdet
    info = torch.empty(1, dtype=torch.int32, device=x.device)
    flags = (0 if lookahead else _lib.POTRF_NO_LOOKAHEAD) | (0 if fused else _lib.POTRF_UNFUSED)
    _lib.check(
        lib.gpar_logpdf_dense(
            ctypes.byref(ck.fspec), ctypes.byref(ck.kspec), x.data_ptr(), n, _ld(x), y.data_ptr(), int(y.stride(0)), nptr, float(jitter),
            z.data_ptr(), _ld(z), A.data_ptr(), _ld(A), words[1:].data_ptr(), info.data_ptr(), words.data_ptr(), flags,
            stream_ptr(x.device),
        ),
        "gpar_logpdf_dense",
    )
    return words[0], words[1:], info, A


def logpdf_dense_grad(ck, x, y, noise_diag, jitter, periodic, lookahead=True, fused=True):
    """One dense layer's log marginal likelihood AND its gradient ingredients in one library call (gpar_logpdf_dense_grad).
    Returns (out, half_diag, info, A): out = [value, logdet, GRAD_NACC moment sums] (device), half_diag = 1/2 diag(W) (device, n),
    info word, the (n + 1) x (n + 1) factor buffer."""
    lib = _lib.load()
    _check_mat(x, "x")
    n, dev = x.shape[0], x.device
    y = y.reshape(-1)
    if y.numel() != n or y.dtype != torch.float64 or not y.is_cuda:
        raise ValueError("y must hold one fp64 device value per row of x")
    nptr = None
    if noise_diag is not None:
        noise_diag = noise_diag.reshape(-1).contiguous()
        if noise_diag.numel() != n:
            raise ValueError("noise_diag must hold one value per row of x")
        nptr = noise_diag.data_ptr()
    dz = max(ck.dz, 1)
    z = alloc_matrix(n, dz, dev)
    zd = alloc_matrix(n, dz, dev, zero=True) if periodic else None
    A = alloc_matrix(n + 1, n + 1, dev)
    X = alloc_matrix(n, n, dev)
    W = alloc_matrix(n, n, dev)
    nt = (n + 63) // 64
    nblocks = max(1, min(nt * (nt + 1) // 2, 1024))
    work = torch.empty(nblocks * _lib.GRAD_NACC + n, dtype=torch.float64, device=dev)   # gradient partials, then alpha
    out = torch.empty(2 + _lib.GRAD_NACC, dtype=torch.float64, device=dev)
    half_diag = torch.empty(n, dtype=torch.float64, device=dev)
    info = torch.empty(1, dtype=torch.int32, device=dev)
    flags = (0 if lookahead else _lib.POTRF_NO_LOOKAHEAD) | (0 if fused else _lib.POTRF_UNFUSED)
    _lib.check(
        lib.gpar_logpdf_dense_grad(
            ctypes.byref(ck.fspec), ctypes.byref(ck.kspec), x.data_ptr(), n, _ld(x), y.data_ptr(), int(y.stride(0)), nptr, float(jitter),
            z.data_ptr(), None if zd is None else zd.data_ptr(), _ld(z), A.data_ptr(), _ld(A), X.data_ptr(), _ld(X), W.data_ptr(), _ld(W),
            work[nblocks * _lib.GRAD_NACC:].data_ptr(), work.data_ptr(), nblocks, out.data_ptr(), half_diag.data_ptr(), info.data_ptr(), flags,
            stream_ptr(dev),
        ),
        "gpar_logpdf_dense_grad",
    )
    return out, half_diag, info, A


def factor_dense_batch(items, jitter, fused=True):
    """Augmented matrices [[k_b(x_b, x_b) + diag(noise_b) + jitter I, .], [y_b^T, 0]] of layers b that share their number of rows,
    built per layer (gpar_logpdf_dense_build) into one buffer and factored in lock-step (gpar_potrf_batch).
    `items`: (compiled kernel, x, y, noise_diag or None) per layer.  Returns (A, logdet, info): A is the (batch (n + 1)) x (n + 1)
    buffer - block b holds L_b, (L_b^-1 y_b)^T in its last row and -|L_b^-1 y_b|^2 in its corner - logdet / info `batch` words."""
    lib = _lib.load()
    batch = len(items)
    x0 = items[0][1]
    n, dev = x0.shape[0], x0.device
    A = alloc_matrix(batch * (n + 1), n + 1, dev)   # matrix b = rows b (n + 1) ... of one buffer
    lda = _ld(A)
    stride = (n + 1) * lda          # lda is a multiple of 16: every matrix starts 128-byte aligned
    logdet = torch.empty(batch, dtype=torch.float64, device=dev)
    info = torch.empty(batch, dtype=torch.int32, device=dev)
    st = stream_ptr(dev)
    keep = []
    for b, (ck, x, y, noise_diag) in enumerate(items):
        _check_mat(x, "x")
        y = y.reshape(-1)
        if x.shape[0] != n or y.numel() != n or y.dtype != torch.float64 or not y.is_cuda:
            raise ValueError("every layer of a batch must hold one fp64 device value per row, and the same number of rows")
        nptr = None
        if noise_diag is not None:
            noise_diag = noise_diag.reshape(-1).contiguous()
            if noise_diag.numel() != n:
                raise ValueError("noise_diag must hold one value per row of x")
            nptr = noise_diag.data_ptr()
        z = alloc_matrix(n, max(ck.dz, 1), dev)
        keep.append((z, y, noise_diag))
        _lib.check(
            lib.gpar_logpdf_dense_build(
                ctypes.byref(ck.fspec), ctypes.byref(ck.kspec), x.data_ptr(), n, _ld(x), y.data_ptr(), int(y.stride(0)), nptr, float(jitter),
                z.data_ptr(), _ld(z), A.data_ptr() + 8 * b * stride, lda, logdet.data_ptr() + 8 * b, info.data_ptr() + 4 * b, st,
            ),
            "gpar_logpdf_dense_build",
        )
    flags = 0 if fused else _lib.POTRF_UNFUSED
    _lib.check(lib.gpar_potrf_batch(A.data_ptr(), batch, stride, n + 1, n, lda, logdet.data_ptr(), info.data_ptr(), flags, st), "gpar_potrf_batch")
    return A, logdet, info


def logpdf_dense_batch(items, jitter, fused=True):
    """log N(y_b; 0, k_b(x_b, x_b) + diag(noise_b) + jitter I) for layers b that share their number of rows and do not feed one
    another: factor_dense_batch, then one gpar_logpdf_dense_finish.  Returns (values, info): `batch` device words each."""
    lib = _lib.load()
    A, logdet, info = factor_dense_batch(items, jitter, fused=fused)
    batch = len(items)
    n = A.shape[1] - 1
    values = torch.empty(batch, dtype=torch.float64, device=A.device)
    _lib.check(lib.gpar_logpdf_dense_finish(A.data_ptr(), batch, (n + 1) * _ld(A), n, _ld(A), logdet.data_ptr(), values.data_ptr(),
                                            stream_ptr(A.device)), "gpar_logpdf_dense_finish")
    return values, info


def logpdf_lockstep(layers, x, y, w, jitter, fused=True):
    """The log marginal likelihoods of layers that share their rows and do not feed one another, and their sum, in ONE library
    call (gpar_logpdf_lockstep).  `layers`: (compiled kernel, noise variance, column of y / w) per layer; x: n x width, the widest
    design matrix ([inputs, y_0 .. y_(p-2)] for a GPAR) - every layer's feature map selects its own columns; y, w: n x p device
    matrices (w None: unit weights).  Returns (values, total, info): `batch` words, one word, `batch` words, all on the device."""
    lib = _lib.load()
    _check_mat(x, "x")
    _check_mat(y, "y")
    batch = len(layers)
    n, dev = x.shape[0], x.device
    if y.dim() != 2 or y.shape[0] != n or (w is not None and (w.shape != y.shape or w.dtype != torch.float64 or not w.is_cuda)):
        raise ValueError("y (and w) must be n x p fp64 device matrices")
    if w is not None and w.stride(1) != 1:
        w = w.contiguous()
    arr = (_lib.Layer * batch)()
    dz = 1
    for b, (ck, noise, col) in enumerate(layers):
        if not 0 <= col < y.shape[1]:
            raise ValueError("observed column out of range")
        arr[b].fs = ctypes.pointer(ck.fspec)
        arr[b].ks = ctypes.pointer(ck.kspec)
        arr[b].noise = float(noise)
        arr[b].y_col = int(col)
        dz = max(dz, ck.dz)
    A = alloc_matrix(batch * (n + 1), n + 1, dev)
    lda = _ld(A)
    z = alloc_matrix(batch * max(n, 1), dz, dev)
    nd = torch.empty(batch * max(n, 1), dtype=torch.float64, device=dev) if w is not None else None
    words = torch.empty(2 * batch + 1, dtype=torch.float64, device=dev)   # logdet[batch], value[batch], total
    info = torch.empty(batch, dtype=torch.int32, device=dev)
    _lib.check(
        lib.gpar_logpdf_lockstep(
            arr, batch, x.data_ptr(), n, _ld(x), y.data_ptr(), _ld(y), None if w is None else w.data_ptr(), 0 if w is None else _ld(w),
            float(jitter), z.data_ptr(), _ld(z), None if nd is None else nd.data_ptr(), A.data_ptr(), lda, (n + 1) * lda,
            words.data_ptr(), info.data_ptr(), words[batch:].data_ptr(), words[2 * batch:].data_ptr(), 0 if fused else _lib.POTRF_UNFUSED,
            stream_ptr(dev),
        ),
        "gpar_logpdf_lockstep",
    )
    return words[batch:2 * batch], words[2 * batch], info


def gram(ck, z1, z2=None, out=None, lower=False, diag_add=None, diag_const=0.0, row_scale=None):
    """K = k(z1, z2) (z2 None: symmetric, optionally lower-only, + diag_add + diag_const on the diagonal); with `row_scale`
    (n1 weights) row a is multiplied by row_scale[a]."""
    lib = _lib.load()
    sym = z2 is None
    if sym:
        z2 = z1
    _check_mat(z1, "z1")
    _check_mat(z2, "z2")
    n1, n2 = z1.shape[0], z2.shape[0]
    if out is None:
        out = alloc_matrix(n1, n2, z1.device)
    _check_mat(out, "out")
    flags = _lib.GRAM_LOWER if (lower and sym) else 0
    dptr = None
    if diag_add is not None:
        if not sym:
            raise ValueError("diag_add requires the symmetric Gram")
        _check_mat(diag_add, "diag_add")
        diag_add = diag_add.contiguous()
        dptr = diag_add.data_ptr()
    rptr = None
    if row_scale is not None:
        if row_scale.dim() != 1 or row_scale.numel() != n1 or row_scale.dtype != torch.float64:
            raise ValueError("row_scale must be a vector of n1 fp64 weights")
        row_scale = row_scale.contiguous()
        rptr = row_scale.data_ptr()
    _lib.check(
        lib.gpar_gram(
            ctypes.byref(ck.kspec), z1.data_ptr(), n1, _ld(z1), z2.data_ptr(), n2, _ld(z2), ck.dz,
            out.data_ptr(), _ld(out), flags, dptr, float(diag_const), rptr, stream_ptr(z1.device),
        ),
        "gpar_gram",
    )
    return out


def gram_batch_(ck, z_all, batch, out, lower=False, diag_add=None, diag_const=0.0):
    """out block b ((batch n) x n, stacked by rows) <- k(z_b, z_b) + diag(diag_add) + diag_const I for the `batch` input sets
    stacked by rows in z_all: one launch (gpar_gram_batch)."""
    lib = _lib.load()
    _check_mat(z_all, "z_all")
    _check_mat(out, "out")
    n = out.shape[1]
    if z_all.shape[0] != batch * n or out.shape[0] != batch * n:
        raise ValueError("shapes of a batched Gram matrix do not match")
    dptr = None
    if diag_add is not None:
        diag_add = diag_add.reshape(-1).contiguous()
        if diag_add.numel() != n:
            raise ValueError("diag_add must hold one value per row")
        dptr = diag_add.data_ptr()
    _lib.check(
        lib.gpar_gram_batch(ctypes.byref(ck.kspec), z_all.data_ptr(), n, _ld(z_all), n * _ld(z_all), ck.dz, out.data_ptr(), _ld(out),
                            n * _ld(out), _lib.GRAM_LOWER if lower else 0, dptr, float(diag_const), batch, stream_ptr(out.device)),
        "gpar_gram_batch",
    )
    return out


def gram_diag(ck, z):
    lib = _lib.load()
    _check_mat(z, "z")
    out = torch.empty(z.shape[0], dtype=torch.float64, device=z.device)
    _lib.check(
        lib.gpar_gram_diag(ctypes.byref(ck.kspec), z.data_ptr(), z.shape[0], _ld(z), ck.dz, out.data_ptr(), stream_ptr(z.device)),
        "gpar_gram_diag",
    )
    return out


def potrf_(A, nf=None, logdet=None, info=None, lookahead=True, fused=True):
    """In-place (partial) Cholesky of the lower triangle of the square matrix A; returns (logdet, info) device
    scalars (logdet accumulates, info is sticky: pass fresh zeros).  `lookahead=False`: the caller has several
    factorisations in flight itself (see gpar_potrf_ex)."""
    lib = _lib.load()
    _check_mat(A, "A")
    N = A.shape[0]
    if A.shape[1] != N:
        raise ValueError("A must be square")
    nf = N if nf is None else int(nf)
    if logdet is None:
        logdet = torch.zeros(1, dtype=torch.float64, device=A.device)
    if info is None:
        info = torch.zeros(1, dtype=torch.int32, device=A.device)
    flags = (0 if lookahead else _lib.POTRF_NO_LOOKAHEAD) | (0 if fused else _lib.POTRF_UNFUSED)
    _lib.check(
        lib.gpar_potrf_ex(A.data_ptr(), N, nf, _ld(A), logdet.data_ptr(), info.data_ptr(), flags, stream_ptr(A.device)), "gpar_potrf_ex"
    )
    return logdet, info


def potrf_batch_(A, batch, nf=None, fused=True):
    """In-place (partial) Cholesky of `batch` square N x N matrices stacked by rows in A ((batch N) x N, one leading
    dimension), factored in lock-step (gpar_potrf_batch); returns (logdet, info): `batch` device words each."""
    lib = _lib.load()
    _check_mat(A, "A")
    N = A.shape[1]
    if A.shape[0] != batch * N:
        raise ValueError("A must hold batch square matrices stacked by rows")
    nf = N if nf is None else int(nf)
    logdet = torch.zeros(batch, dtype=torch.float64, device=A.device)
    info = torch.zeros(batch, dtype=torch.int32, device=A.device)
    _lib.check(
        lib.gpar_potrf_batch(A.data_ptr(), batch, N * _ld(A), N, nf, _ld(A), logdet.data_ptr(), info.data_ptr(),
                             0 if fused else _lib.POTRF_UNFUSED, stream_ptr(A.device)),
        "gpar_potrf_batch",
    )
    return logdet, info


def trsm_rlt_(L, B, when=None):
    """B <- B L^-T (rows of B solved by forward substitution).  `when` = (flag, sense): predicated on the device word flag[0] - the
    solve happens iff (flag[0] != 0) == sense (gpar_trsm_rlt_if); B is left untouched otherwise."""
    lib = _lib.load()
    _check_mat(L, "L")
    _check_mat(B, "B")
    n = L.shape[0]
    if B.shape[1] != n:
        raise ValueError("B must have as many columns as L has rows")
    if when is not None:
        flag, sense = when
        if flag.dtype != torch.int32 or not flag.is_cuda:
            raise TypeError("the predicate of a solve is an int32 device word")
        _lib.check(lib.gpar_trsm_rlt_if(L.data_ptr(), n, _ld(L), B.data_ptr(), B.shape[0], _ld(B), flag.data_ptr(), int(bool(sense)),
                                        stream_ptr(B.device)), "gpar_trsm_rlt_if")
        return B
    _lib.check(lib.gpar_trsm_rlt(L.data_ptr(), n, _ld(L), B.data_ptr(), B.shape[0], _ld(B), stream_ptr(B.device)), "gpar_trsm_rlt")
    return B


def vfe_factor(G, c, ys, kdiag, d, diag_add, with_trace, lookahead=True, fused=True):
    """The factor of A = I-shifted G with the row c appended, and the inducing-point bound from it (gpar_vfe_assemble, gpar_potrf,
    gpar_vfe_value): returns (A buffer (M + 1) x (M + 1), logdet word, info word, bound as a 0-d device tensor)."""
    lib = _lib.load()
    _check_mat(G, "G")
    M, n, dev = G.shape[0], ys.numel(), G.device
    c, ys, kdiag, d = (t.reshape(-1).contiguous() for t in (c, ys, kdiag, d))
    A = alloc_matrix(M + 1, M + 1, dev)
    words = torch.empty(258, dtype=torch.float64, device=dev)   # logdet, bound, then 64 x 4 partial sums
    info = torch.empty(1, dtype=torch.int32, device=dev)
    st = stream_ptr(dev)
    _lib.check(lib.gpar_vfe_assemble(G.data_ptr(), M, _ld(G), c.data_ptr(), ys.data_ptr(), kdiag.data_ptr(), d.data_ptr(), n, float(diag_add),
                                     A.data_ptr(), _ld(A), words[2:].data_ptr(), words.data_ptr(), info.data_ptr(), st), "gpar_vfe_assemble")
    flags = (0 if lookahead else _lib.POTRF_NO_LOOKAHEAD) | (0 if fused else _lib.POTRF_UNFUSED)
    _lib.check(lib.gpar_potrf_ex(A.data_ptr(), M + 1, M, _ld(A), words.data_ptr(), info.data_ptr(), flags, st), "gpar_potrf_ex")
    _lib.check(lib.gpar_vfe_value(words[2:].data_ptr(), words.data_ptr(), A.data_ptr(), _ld(A), M, n, int(bool(with_trace)), words[1:].data_ptr(), st),
               "gpar_vfe_value")
    return A, words[0:1], info, words[1]


def chol_spread(L, limit):
    """(spread, flag) device words of a Cholesky factor: (max L_jj / min L_jj)^2 and whether it exceeds `limit` (gpar_chol_spread)."""
    lib = _lib.load()
    _check_mat(L, "L")
    spread = torch.empty(1, dtype=torch.float64, device=L.device)
    flag = torch.empty(1, dtype=torch.int32, device=L.device)
    _lib.check(lib.gpar_chol_spread(L.data_ptr(), L.shape[0], _ld(L), float(limit), spread.data_ptr(), flag.data_ptr(), stream_ptr(L.device)),
               "gpar_chol_spread")
    return spread, flag


def trsm_rln_(L, B):
    """B <- B L^-1 (rows of B solved by backward substitution)."""
    lib = _lib.load()
    _check_mat(L, "L")
    _check_mat(B, "B")
    n = L.shape[0]
    if B.shape[1] != n:
        raise ValueError("B must have as many columns as L has rows")
    _lib.check(lib.gpar_trsm_rln(L.data_ptr(), n, _ld(L), B.data_ptr(), B.shape[0], _ld(B), stream_ptr(B.device)), "gpar_trsm_rln")
    return B


def gemm(A, B, ta=False, tb=False, alpha=1.0, beta=0.0, out=None, c_lower=False, a_lower=False, k_from_row=False, k_to_col=False):
    """out <- alpha op(A) op(B) + beta out;  ta: A stored k x m;  tb: B stored n x k.  `k_from_row` / `k_to_col`: op(A) is
    upper triangular (stored zeros left of its diagonal) / op(B) is upper triangular (stored zeros below it): the K loop
    skips the blocks that are zero."""
    lib = _lib.load()
    _check_mat(A, "A")
    _check_mat(B, "B")
    m, k = (A.shape[1], A.shape[0]) if ta else (A.shape[0], A.shape[1])
    n, kb = (B.shape[0], B.shape[1]) if tb else (B.shape[1], B.shape[0])
    if k != kb:
        raise ValueError(f"inner dimensions differ: {k} vs {kb}")
    if out is None:
        if (n == 1 and tb and not ta and beta == 0.0 and not (c_lower or a_lower or k_from_row or k_to_col) and B.dim() == 2 and m > 0
                and os.environ.get("GPAR_GEMV", "1") != "0"):
            # one column: a matrix-vector product (gpar_gemv: one wave per row) instead of 128-wide tiles with one live column
            y = torch.empty(m, 1, dtype=torch.float64, device=A.device)
            _lib.check(lib.gpar_gemv(A.data_ptr(), m, k, _ld(A), B.data_ptr(), 1, float(alpha), y.data_ptr(), 1, stream_ptr(A.device)), "gpar_gemv")
            return y
        out = alloc_matrix(m, n, A.device)
        if beta != 0.0:
            raise ValueError("beta != 0 needs an `out`")
    _check_mat(out, "out")
    flags = ((_lib.GEMM_C_LOWER if c_lower else 0) | (_lib.GEMM_A_LOWER if a_lower else 0) | (_lib.GEMM_K_FROM_ROW if k_from_row else 0)
             | (_lib.GEMM_K_TO_COL if k_to_col else 0))
    # few output tiles but a very long K: cut K into slices so that the launch fills the chip's 512 workgroup slots (two
    # 73.7 KB workgroups per CU) in one round; deterministic two-pass sum.  (The workspace comes from torch's stream-aware
    # caching allocator: no hipMalloc after the first call of a given size.)
    tm, tn = (m + 127) // 128, (n + 127) // 128
    tiles = (min(tm, tn) * (min(tm, tn) + 1) // 2 + (tm - min(tm, tn)) * min(tm, tn)) if c_lower else tm * tn
    if not (a_lower or k_from_row or k_to_col) and k >= 8192 and tiles <= 128:
        splits = max(2, min(64, 512 // max(tiles, 1), k // 1024))
        work = torch.empty(lib.gpar_workspace_doubles(_lib.WS_GEMM_SPLITK, m, n, splits), dtype=torch.float64, device=A.device)
        _lib.check(
            lib.gpar_gemm_splitk(
                int(ta), int(tb), m, n, k, float(alpha), A.data_ptr(), _ld(A), B.data_ptr(), _ld(B), float(beta),
                out.data_ptr(), _ld(out), flags, splits, work.data_ptr(), stream_ptr(A.device),
            ),
            "gpar_gemm_splitk",
        )
        return out
    _lib.check(
        lib.gpar_gemm(
            int(ta), int(tb), m, n, k, float(alpha), A.data_ptr(), _ld(A), B.data_ptr(), _ld(B), float(beta),
            out.data_ptr(), _ld(out), flags, stream_ptr(A.device),
        ),
        "gpar_gemm",
    )
    return out


def dot(x, incx, y, incy, n, out=None, accumulate=False):
    lib = _lib.load()
    if out is None:
        out = torch.zeros(1, dtype=torch.float64, device=x.device)
    _lib.check(
        lib.gpar_dot(x.data_ptr(), int(incx), y.data_ptr(), int(incy), int(n), out.data_ptr(), int(accumulate), stream_ptr(x.device)),
        "gpar_dot",
    )
    return out


def gemv_t(A, v):
    """A^T v for a tall matrix A (rows x cols) and a vector v (rows): new vector of `cols` entries."""
    lib = _lib.load()
    _check_mat(A, "A")
    rows, cols = A.shape
    v = v.reshape(-1).contiguous()
    if v.numel() != rows or v.dtype != torch.float64:
        raise ValueError("v must hold one fp64 weight per row of A")
    out = torch.empty(cols, dtype=torch.float64, device=A.device)
    work = torch.empty(max(1, lib.gpar_workspace_doubles(_lib.WS_GEMV_T, rows, cols, 0)), dtype=torch.float64, device=A.device)
    _lib.check(lib.gpar_gemv_t(A.data_ptr(), rows, cols, _ld(A), v.data_ptr(), out.data_ptr(), work.data_ptr(), stream_ptr(A.device)),
               "gpar_gemv_t")
    return out


def rownorm2(A):
    """Squared Euclidean norms of the rows of A: new vector."""
    lib = _lib.load()
    _check_mat(A, "A")
    out = torch.empty(A.shape[0], dtype=torch.float64, device=A.device)
    _lib.check(lib.gpar_rownorm2(A.data_ptr(), A.shape[0], A.shape[1], _ld(A), out.data_ptr(), stream_ptr(A.device)), "gpar_rownorm2")
    return out


def pack_lower(A, out=None):
    """Lower triangle (diagonal included) of the square matrix A as a contiguous vector of n (n + 1) / 2 doubles."""
    lib = _lib.load()
    _check_mat(A, "A")
    n = A.shape[0]
    if out is None:
        out = torch.empty(n * (n + 1) // 2, dtype=torch.float64, device=A.device)
    _lib.check(lib.gpar_pack_lower(A.data_ptr(), n, _ld(A), out.data_ptr(), stream_ptr(A.device)), "gpar_pack_lower")
    return out


def unpack_lower_(packed, A):
    """Inverse of pack_lower into the lower triangle of A (the strict upper triangle is left alone)."""
    lib = _lib.load()
    _check_mat(A, "A")
    n = A.shape[0]
    if packed.numel() < n * (n + 1) // 2 or not packed.is_contiguous():
        raise ValueError("packed buffer too small or not contiguous")
    _lib.check(lib.gpar_unpack_lower(packed.data_ptr(), n, A.data_ptr(), _ld(A), stream_ptr(A.device)), "gpar_unpack_lower")
    return A


def randn(seed, offset, rows, cols, device):
    lib = _lib.load()
    out = alloc_matrix(rows, cols, device)
    _lib.check(
        lib.gpar_randn(int(seed) & (2**64 - 1), int(offset) & (2**64 - 1), out.data_ptr(), rows, cols, _ld(out), stream_ptr(device)),
        "gpar_randn",
    )
    return out


def gemm_batch_(A, B, out, batch, ta=False, tb=False, alpha=1.0, beta=0.0, c_lower=False):
    """out_b <- alpha op(A_b) op(B_b) + beta out_b for `batch` problems of one shape whose operands are stacked by rows in A, B
    and out (block b = rows [b R, (b + 1) R) with R = rows / batch): one launch (gpar_gemm_batch)."""
    lib = _lib.load()
    for name, t in (("A", A), ("B", B), ("out", out)):
        _check_mat(t, name)
        if t.shape[0] % batch:
            raise ValueError(f"{name}: rows must be a multiple of the batch size")
    ra, rb, rc = A.shape[0] // batch, B.shape[0] // batch, out.shape[0] // batch
    m, k = (A.shape[1], ra) if ta else (ra, A.shape[1])
    n, kb = (rb, B.shape[1]) if tb else (B.shape[1], rb)
    if k != kb or rc != m or out.shape[1] != n:
        raise ValueError("shapes of a batched product do not match")
    _lib.check(
        lib.gpar_gemm_batch(int(ta), int(tb), m, n, k, float(alpha), A.data_ptr(), _ld(A), ra * _ld(A), B.data_ptr(), _ld(B), rb * _ld(B),
                            float(beta), out.data_ptr(), _ld(out), rc * _ld(out), _lib.GEMM_C_LOWER if c_lower else 0, batch,
                            stream_ptr(out.device)),
        "gpar_gemm_batch",
    )
    return out


def trmv_lower(L, x):
    """L x for the lower triangle of the square matrix L and one column x (n x 1, any row stride); new n x 1 tensor."""
    lib = _lib.load()
    _check_mat(L, "L")
    n = L.shape[0]
    if x.dim() != 2 or x.shape[0] != n or x.shape[1] != 1 or x.dtype != torch.float64:
        raise ValueError("x must be an n x 1 fp64 matrix")
    y = torch.empty(n, 1, dtype=torch.float64, device=L.device)
    _lib.check(lib.gpar_trmv_lower(L.data_ptr(), n, _ld(L), x.data_ptr(), int(x.stride(0)), y.data_ptr(), 1, stream_ptr(L.device)),
               "gpar_trmv_lower")
    return y


def trmv_lower_batch_(Ls, batch, X, out, add=None):
    """Column b of `out` (n x batch) <- L_b X[:, b] (+ add[b n : (b + 1) n]) for the `batch` lower-triangular n x n matrices
    stacked by rows in Ls; X: n x batch (any strides), add: (batch n) x 1 or None.  One launch."""
    lib = _lib.load()
    _check_mat(Ls, "Ls")
    n = Ls.shape[1]
    if Ls.shape[0] != batch * n or X.shape != (n, batch) or out.shape != (n, batch) or X.dtype != torch.float64 or out.dtype != torch.float64:
        raise ValueError("shapes of a batched triangular product do not match")
    aptr, inca, stride_add = None, 0, 0
    if add is not None:
        if add.numel() != batch * n or add.dtype != torch.float64:
            raise ValueError("add must hold one value per row and matrix")
        add = add.reshape(batch * n, -1)
        aptr, inca, stride_add = add.data_ptr(), int(add.stride(0)), n * int(add.stride(0))
    _lib.check(
        lib.gpar_trmv_lower_batch(Ls.data_ptr(), batch, n * _ld(Ls), n, _ld(Ls), X.data_ptr(), int(X.stride(0)), int(X.stride(1)), aptr, inca,
                                  stride_add, out.data_ptr(), int(out.stride(0)), int(out.stride(1)), stream_ptr(Ls.device)),
        "gpar_trmv_lower_batch",
    )
    return out


def percentile_index(num, q):
    """(k, g) such that numpy's default ("linear", Hyndman & Fan 7) percentile q of `num` sorted values v is
    lerp(v[k], v[k + 1], g), with virtual index h = (num - 1) * (q / 100), k = floor(h), g = h - k - the expression
    numpy >= 2.0 evaluates, so the device result is bit-identical to np.percentile there."""
    virtual = (num - 1) * (float(q) / 100.0)
    k = int(math.floor(virtual))
    g = virtual - k
    if k < 0:
        k, g = 0, 0.0
    if k >= num - 1:
        k, g = num - 1, 0.0
    return k, g


def sample_stats(samples, q_lo=None, q_hi=None):
    """samples: contiguous (S, ...) fp64 device tensor.  Returns (mean, lo, hi) over axis 0 (lo / hi None unless both
    percentiles are given) - np.mean / np.percentile(method="linear") of the reference's predict, on the device."""
    if samples.dtype != torch.float64 or not samples.is_contiguous() or samples.dim() < 2:
        raise ValueError("samples must be a contiguous fp64 tensor of shape (S, ...)")
    lib = _lib.load()
    S = samples.shape[0]
    count = samples[0].numel()
    mean = torch.empty(samples.shape[1:], dtype=torch.float64, device=samples.device)
    want = q_lo is not None and q_hi is not None
    lo = torch.empty_like(mean) if want else None
    hi = torch.empty_like(mean) if want else None
    (k0, g0), (k1, g1) = (percentile_index(S, q_lo), percentile_index(S, q_hi)) if want else ((0, 0.0), (0, 0.0))
    _lib.check(
        lib.gpar_sample_stats(samples.data_ptr(), int(S), int(count), int(count), k0, g0, k1, g1, mean.data_ptr(),
                              lo.data_ptr() if want else None, hi.data_ptr() if want else None, stream_ptr(samples.device)),
        "gpar_sample_stats",
    )
    return mean, lo, hi


def featurize_dfreq(ck, x):
    """zd = d features / d freq (zero for non-periodic features)."""
    _check_mat(x, "x")
    lib = _lib.load()
    n = x.shape[0]
    zd = alloc_matrix(n, max(ck.dz, 1), x.device, zero=True)
    if ck.dz:
        _lib.check(
            lib.gpar_featurize_dfreq(ctypes.byref(ck.fspec), x.data_ptr(), n, _ld(x), zd.data_ptr(), _ld(zd), stream_ptr(x.device)),
            "gpar_featurize_dfreq",
        )
    return zd


def gram_grad(ck, z, zd, W, nblocks=None):
    """Raw gradient moment sums (length GRAD_NACC, device tensor) of 1/2 sum W dK/dtheta; see csrc/gram.h."""
    lib = _lib.load()
    _check_mat(z, "z")
    _check_mat(W, "W")
    n = z.shape[0]
    nt = (n + 63) // 64
    ntiles = nt * (nt + 1) // 2
    if nblocks is None:
        nblocks = max(1, min(ntiles, 1024))
    work = torch.empty(nblocks * _lib.GRAD_NACC, dtype=torch.float64, device=z.device)
    out = torch.empty(_lib.GRAD_NACC, dtype=torch.float64, device=z.device)
    zdp = None
    if zd is not None:
        _check_mat(zd, "zd")
        if _ld(zd) != _ld(z):
            raise ValueError("z and zd must share a leading dimension")
        zdp = zd.data_ptr()
    _lib.check(
        lib.gpar_gram_grad(
            ctypes.byref(ck.kspec), z.data_ptr(), zdp, n, _ld(z), ck.dz, W.data_ptr(), _ld(W), work.data_ptr(), nblocks,
            out.data_ptr(), stream_ptr(z.device),
        ),
        "gpar_gram_grad",
    )
    return out


GRAD_SYM, GRAD_RECT, GRAD_DIAG = 0, 1, 2


def gram_grad_cross(ck, z1, zd1, z2, zd2, W, mode, nblocks=None):
    """Raw moment sums (length GRAD_NACC, device tensor) of sum W dK/dtheta for the weight shapes of the VFE bound:
    GRAD_SYM (z2 is z1, W symmetric, lower triangle read), GRAD_RECT (W: n1 x n2), GRAD_DIAG (z2 is z1, W: n1 weights of
    the pairs (a, a)).  Full sums: no factor 1/2."""
    lib = _lib.load()
    _check_mat(z1, "z1")
    _check_mat(z2, "z2")
    n1, n2 = z1.shape[0], z2.shape[0]
    nt1, nt2 = (n1 + 63) // 64, (n2 + 63) // 64
    ntiles = nt1 * (nt1 + 1) // 2 if mode == GRAD_SYM else (nt1 * nt2 if mode == GRAD_RECT else nt1)
    if nblocks is None:
        nblocks = max(1, min(ntiles, 1024))
    if mode == GRAD_DIAG:
        if W.dim() != 1 or not W.is_contiguous() or W.numel() != n1:
            raise ValueError("GRAD_DIAG takes a contiguous vector of n1 weights")
        ldw = 1
    else:
        _check_mat(W, "W")
        ldw = _ld(W)
    work = torch.empty(nblocks * _lib.GRAD_NACC, dtype=torch.float64, device=z1.device)
    out = torch.empty(_lib.GRAD_NACC, dtype=torch.float64, device=z1.device)
    for a, b in ((zd1, z1), (zd2, z2)):
        if a is not None and _ld(a) != _ld(b):
            raise ValueError("features and their frequency derivatives must share a leading dimension")
    _lib.check(
        lib.gpar_gram_grad_cross(
            ctypes.byref(ck.kspec), z1.data_ptr(), None if zd1 is None else zd1.data_ptr(), n1, _ld(z1), z2.data_ptr(),
            None if zd2 is None else zd2.data_ptr(), n2, _ld(z2), ck.dz, W.data_ptr(), ldw, int(mode), work.data_ptr(), nblocks,
            out.data_ptr(), stream_ptr(z1.device),
        ),
        "gpar_gram_grad_cross",
    )
    return out


def gram_input_grad(ck, z1, z2, W, mode):
    """out[a][q] = sum_b W(a, b) d k(z1_a, z2_b) / d z1_a[q] (feature space; n1 x dz).  mode GRAD_RECT (W: n1 x n2) or
    GRAD_SYM (z2 is z1, W symmetric, lower triangle read)."""
    lib = _lib.load()
    _check_mat(z1, "z1")
    _check_mat(z2, "z2")
    _check_mat(W, "W")
    n1, n2 = z1.shape[0], z2.shape[0]
    out = alloc_matrix(n1, max(ck.dz, 1), z1.device, zero=True)
    if ck.dz == 0 or n1 == 0:
        return out
    row_blocks = (n1 + 63) // 64
    col_tiles = max(1, (n2 + 63) // 64)
    nsplit = max(1, min(col_tiles, 512 // max(row_blocks, 1)))
    work = torch.empty(max(1, lib.gpar_workspace_doubles(_lib.WS_INPUT_GRAD, n1, ck.dz, nsplit)), dtype=torch.float64, device=z1.device)
    _lib.check(
        lib.gpar_gram_input_grad(ctypes.byref(ck.kspec), z1.data_ptr(), n1, _ld(z1), z2.data_ptr(), n2, _ld(z2), ck.dz, W.data_ptr(),
                                 _ld(W), int(mode), nsplit, work.data_ptr(), out.data_ptr(), _ld(out), stream_ptr(z1.device)),
        "gpar_gram_input_grad",
    )
    return out


def chol_inverse(L, with_x=False):
    """Lower triangle of (L L^T)^-1 (new matrix) from the Cholesky factor L; triangular-aware (2 n^3 / 3 flops).  `with_x`: also the
    workspace the call leaves behind, X = L^-T in its upper triangle (the strict lower triangle is scratch)."""
    lib = _lib.load()
    _check_mat(L, "L")
    n = L.shape[0]
    X = alloc_matrix(n, n, L.device)
    Kinv = alloc_matrix(n, n, L.device)
    _lib.check(
        lib.gpar_chol_inverse(L.data_ptr(), n, _ld(L), X.data_ptr(), _ld(X), Kinv.data_ptr(), _ld(Kinv), stream_ptr(L.device)),
        "gpar_chol_inverse",
    )
    return (Kinv, X) if with_x else Kinv


def trmv_upper(U, x):
    """U x for an upper-triangular U (the strict lower triangle is not read) and a single vector x (any shape with n entries, unit or
    row stride): new contiguous vector of n entries (gpar_trmv_upper)."""
    lib = _lib.load()
    _check_mat(U, "U")
    n = U.shape[0]
    x = x.reshape(-1)
    if x.numel() != n or x.dtype != torch.float64 or not x.is_cuda:
        raise ValueError("x must hold one fp64 device value per row of U")
    out = torch.empty(n, dtype=torch.float64, device=U.device)
    _lib.check(lib.gpar_trmv_upper(U.data_ptr(), n, _ld(U), x.data_ptr(), int(x.stride(0)), out.data_ptr(), 1, stream_ptr(U.device)),
               "gpar_trmv_upper")
    return out
