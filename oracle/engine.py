"""Oracle (TEST INFRASTRUCTURE): numpy/scipy fp64 implementation of the engine seam (gpar_amd/engine.py).

It lets the test-suite (i) run the host orchestration (gpar_amd.model / regression / gp) without a GPU, so the
reference's behavioural tests can be ported to CPU, and (ii) compare every HIP primitive and every end-to-end
quantity against an independent fp64 computation.  It is installed explicitly with
`gpar_amd.engine.set_engine(OracleEngine())` by tests, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` only; the product never imports this module and never falls back to it.

Algorithms restated: LAPACK dpotrf / dtrtrs semantics via scipy.linalg (the reference reaches the same routines
through matrix -> lab -> torch.linalg.cholesky / solve_triangular), partial Cholesky =
Schur complement, kernels as in oracle/kernels.py.
"""
import numpy as np
import scipy.linalg
import torch

from . import kernels as ok
from . import philox

__all__ = ["OracleEngine"]


class _Compiled:
    def __init__(self, kernel, width):
        self.kernel = kernel.resolve(width)
        self.width = width
        self.spec = ok.spec_to_dict(self.kernel)


from gpar_amd.engine import NotPositiveDefiniteError as _ProductNotPD  # the exception type the host code catches


class OracleNotPositiveDefinite(_ProductNotPD):
    pass


def _np(t):
    return t.detach().numpy() if isinstance(t, torch.Tensor) else np.asarray(t, dtype=np.float64)



class OracleEngine:
    name = "oracle"

    def __init__(self, seed=0, epsilon=1e-12):
        self.device = torch.device("cpu")
        self.epsilon = float(epsilon)
        self.cholesky_retry_factor = 1.0
        self._seed = int(seed)
        self._calls = 0

    # ---- memory ----------------------------------------------------------------------------------
    def tensor(self, x):
        if isinstance(x, torch.Tensor):
            if x.requires_grad and torch.is_grad_enabled():
                return x.to(device="cpu", dtype=torch.float64)
            return x.detach().to(device="cpu", dtype=torch.float64)
        return torch.as_tensor(np.asarray(x, dtype=np.float64))

    def new_matrix(self, rows, cols, zero=False):
        # NaN-filled unless zeroed: anything the host code reads without having written shows up in tests
        return torch.zeros(rows, cols, dtype=torch.float64) if zero else torch.full((rows, cols), float("nan"), dtype=torch.float64)

    # ---- kernels ---------------------------------------------------------------------------------
    def compile(self, kernel, width):
        return _Compiled(kernel, width)

    def features(self, ck, x):
        return x  # the oracle evaluates kernels on the raw design matrix

    def gram(self, ck, z1, z2=None, lower=False, diag_add=None, diag_const=0.0, out=None, row_scale=None):
        K = ok.gram(ck.spec, _np(z1), None if z2 is None else _np(z2),
                    noise_diag=None if diag_add is None else _np(diag_add), jitter=diag_const)
        if row_scale is not None:
            K = K * _np(row_scale).reshape(-1, 1)
        if out is None:
            return torch.from_numpy(K)
        o = out.numpy()
        if lower and z2 is None:
            il = np.tril_indices(K.shape[0])
            o[il] = K[il]
        else:
            o[...] = K
        return out

    def gram_diag(self, ck, z):
        return torch.from_numpy(ok.gram_diag(ck.spec, _np(z)))

    def kernel_grads(self, ck, x, W):
        """1/2 sum_ab W_ab dK_ab/dtheta for every kernel parameter (W given by its lower triangle)."""
        w = np.tril(_np(W))
        w = w + np.tril(w, -1).T
        return ok.kernel_grads(ck.spec, _np(x), w)

    def kernel_input_grads(self, ck, x1, x2, W, sym=False):
        if sym:
            w = np.tril(_np(W))
            w = w + np.tril(w, -1).T
            return torch.from_numpy(2.0 * ok.kernel_input_grads(ck.spec, _np(x1), _np(x1), w))
        return torch.from_numpy(ok.kernel_input_grads(ck.spec, _np(x1), _np(x2), _np(W)))

    def kernel_diag_input_grads(self, ck, x, w):
        """d / d x of sum_a w_a k(x_a, x_a), by central differences of the kernel diagonal (exact to ~1e-9: the diagonal is a
        polynomial in x for the kernels of this package)."""
        xs, ws = _np(x), _np(w).reshape(-1)
        out = np.zeros_like(xs)
        for c in range(xs.shape[1]):
            step = np.zeros_like(xs)
            step[:, c] = 1e-5
            out[:, c] = ws * (ok.gram_diag(ck.spec, xs + step) - ok.gram_diag(ck.spec, xs - step)) / 2e-5
        return torch.from_numpy(out)

    def kernel_grads_weighted(self, ck, x1, x2, W, sym=False):
        """sum_ab W_ab dK(x1_a, x2_b)/dtheta for every kernel parameter (full sums).  sym: x2 is x1, W symmetric (lower)."""
        if sym:
            w = np.tril(_np(W))
            w = w + np.tril(w, -1).T
            half = ok.kernel_grads(ck.spec, _np(x1), w)
        else:
            a, b = _np(x1), _np(x2)
            n1, n2 = a.shape[0], b.shape[0]
            big = np.zeros((n1 + n2, n1 + n2))
            big[:n1, n1:] = 0.5 * _np(W)
            big[n1:, :n1] = 0.5 * _np(W).T
            half = ok.kernel_grads(ck.spec, np.concatenate([a, b], axis=0), big)
        out = {"coef": [2.0 * c for c in half["coef"]], "factors": []}
        for fgrads in half["factors"]:
            out["factors"].append([{k: (None if v is None else 2.0 * np.asarray(v)) for k, v in g.items()} for g in fgrads])
        return out

    def kernel_grads_vfe(self, ck, x, z, W_fu, W_uu, wdiag):
        """sum_aj W_fu[a, j] dK(x_a, z_j) + sum_ij W_uu[i, j] dK(z_i, z_j) + sum_a wdiag[a] dk(x_a, x_a) for every kernel
        parameter, through the explicit derivative matrices of the symmetric routine on the stacked points [z; x]:
        each cross pair appears twice in a symmetric sum, hence the halves, and the routine returns half the sum."""
        xz = np.concatenate([_np(z), _np(x)], axis=0)
        M, n = _np(z).shape[0], _np(x).shape[0]
        big = np.zeros((M + n, M + n))
        wuu = _np(W_uu)
        big[:M, :M] = 0.5 * (wuu + wuu.T)
        big[M:, :M] = 0.5 * _np(W_fu)
        big[:M, M:] = 0.5 * _np(W_fu).T
        big[M:, M:] = np.diag(_np(wdiag).reshape(-1))
        half = ok.kernel_grads(ck.spec, xz, big)
        out = {"coef": [2.0 * c for c in half["coef"]], "factors": []}
        for fgrads in half["factors"]:
            out["factors"].append([{k: (None if v is None else 2.0 * np.asarray(v)) for k, v in g.items()} for g in fgrads])
        return out

    def kernel_grads_diag(self, ck, x, wdiag):
        """sum_a wdiag[a] dk(x_a, x_a) for every kernel parameter (the symmetric routine returns half the sum)."""
        half = ok.kernel_grads(ck.spec, _np(x), np.diag(_np(wdiag).reshape(-1)))
        out = {"coef": [2.0 * c for c in half["coef"]], "factors": []}
        for fgrads in half["factors"]:
            out["factors"].append([{k: (None if v is None else 2.0 * np.asarray(v)) for k, v in g.items()} for g in fgrads])
        return out

    # ---- factorisations ----------------------------------------------------------------------------
    def potrf_(self, A, nf=None):
        a = A.numpy()
        N = a.shape[0]
        nf = N if nf is None else int(nf)
        low = np.tril(a)
        full = low + np.tril(low, -1).T
        logdet = torch.zeros(1, dtype=torch.float64)
        info = torch.zeros(1, dtype=torch.int32)
        if nf == 0:
            return logdet, info
        try:
            L11 = np.linalg.cholesky(full[:nf, :nf])
        except np.linalg.LinAlgError:
            # locate the first failing pivot as LAPACK would report it
            k = 1
            while k <= nf:
                try:
                    np.linalg.cholesky(full[:k, :k])
                except np.linalg.LinAlgError:
                    break
                k += 1
            info[0] = k
            return logdet, info
        logdet[0] = 2.0 * np.sum(np.log(np.diag(L11)))
        il = np.tril_indices(nf)
        a[:nf, :nf][il] = L11[il]
        if nf < N:
            L21 = scipy.linalg.solve_triangular(L11, full[:nf, nf:], lower=True).T
            a[nf:, :nf] = L21
            S = full[nf:, nf:] - L21 @ L21.T
            il2 = np.tril_indices(N - nf)
            a[nf:, nf:][il2] = S[il2]
        return logdet, info

    def trsm_rlt_(self, L, B):
        if B.shape[0] and B.shape[1]:
            b = B.numpy()
            b[...] = scipy.linalg.solve_triangular(np.tril(L.numpy()), b.T, lower=True).T
        return B

    def trsm_rln_(self, L, B):
        if B.shape[0] and B.shape[1]:
            b = B.numpy()
            b[...] = scipy.linalg.solve_triangular(np.tril(L.numpy()), b.T, lower=True, trans="T").T
        return B

    def chol_inverse(self, L):
        """Lower triangle of (L L^T)^-1 (dense inverse of the explicit product: an independent route)."""
        l = np.tril(_np(L))
        inv = np.linalg.inv(l @ l.T)
        out = torch.full(inv.shape, float("nan"), dtype=torch.float64)
        il = np.tril_indices(inv.shape[0])
        out.numpy()[il] = inv[il]
        return out

    def gemm(self, A, B, ta=False, tb=False, alpha=1.0, beta=0.0, out=None, c_lower=False, a_lower=False):
        a, b = _np(A), _np(B)
        opa = a.T if ta else a
        if a_lower:
            opa = np.tril(opa)
        opb = b.T if tb else b
        prod = alpha * (opa @ opb)
        if out is None:
            return torch.from_numpy(np.ascontiguousarray(prod))
        o = out.numpy()
        if c_lower:
            il = np.tril_indices(prod.shape[0], 0, prod.shape[1])
            o[il] = prod[il] + (beta * o[il] if beta != 0.0 else 0.0)
        else:
            o[...] = prod + (beta * o if beta != 0.0 else 0.0)
        return out

    def gemv_t(self, A, v):
        return torch.from_numpy(_np(A).T @ _np(v).reshape(-1))

    def pack_lower(self, A, out=None):
        a = _np(A)
        il = np.tril_indices(a.shape[0])
        packed = torch.from_numpy(np.ascontiguousarray(a[il]))
        if out is not None:
            out[: packed.numel()] = packed
            return out
        return packed

    def unpack_lower_(self, packed, A):
        a = A.numpy()
        il = np.tril_indices(a.shape[0])
        a[il] = _np(packed)[: len(il[0])]
        return A

    def rownorm2(self, A):
        a = _np(A)
        return torch.from_numpy(np.sum(a * a, axis=1))

    # ---- randomness ------------------------------------------------------------------------------
    def seed(self, seed):
        self._seed = int(seed)
        self._calls = 0

    def randn(self, rows, cols):
        out = torch.from_numpy(philox.randn(self._seed, self._calls, rows, cols))
        self._calls += 1
        return out

    def trmv_lower(self, L, x):
        return torch.from_numpy(np.tril(_np(L)) @ _np(x))

    def sample_stats(self, samples, q_lo=None, q_hi=None):
        """numpy, exactly as the reference does it (regression.py:589-595)."""
        arr = _np(samples)
        mean = torch.from_numpy(np.mean(arr, axis=0))
        if q_lo is None or q_hi is None:
            return mean, None, None
        return mean, torch.from_numpy(np.percentile(arr, q_lo, axis=0)), torch.from_numpy(np.percentile(arr, q_hi, axis=0))

    def pipeline(self, depth=None, rows=None):
        return None  # one CPU, nothing to overlap

    _deferred = None

    def defer_checks(self):
        import contextlib

        return contextlib.nullcontext()

    @staticmethod
    def check_info(info):
        code = int(info.item())
        if code != 0:
            raise OracleNotPositiveDefinite(code)

    _raise_for = check_info
