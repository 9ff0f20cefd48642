"""GPU parity of the C-ABI primitives against numpy/scipy fp64 (oracle) on seeded inputs.

Tolerances (fp64): GEMM-type results are compared with rtol 1e-12 scaled by the magnitude of the summands
(different summation order than numpy's BLAS); Cholesky factors / solves of well-conditioned matrices with
rtol 1e-10; Philox uniforms are bit-exact so normals agree to ~1e-14.
"""
import os

import numpy as np
import pytest
import scipy.linalg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    from gpar_amd import hip

    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")

    def to_dev(a, pad=True):
        a = np.asarray(a, dtype=np.float64)
        if a.ndim == 1:
            return torch.tensor(a, dtype=torch.float64, device=dev)
        if pad:
            out = hip.alloc_matrix(a.shape[0], a.shape[1], dev)
            out.copy_(torch.tensor(a, dtype=torch.float64))
            return out
        return torch.tensor(a, dtype=torch.float64, device=dev)

    return torch, hip, dev, to_dev


def _spd(rng, n, cond=1e3):
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    d = np.exp(rng.uniform(0, np.log(cond), n))
    return (q * d) @ q.T


@pytest.mark.parametrize("ta", [False, True])
@pytest.mark.parametrize("tb", [False, True])
@pytest.mark.parametrize("mnk", [(1, 1, 1), (5, 7, 3), (128, 128, 16), (130, 257, 33), (300, 64, 200), (64, 515, 129),
                                 # K a multiple of 16 with tiles overhanging m and n: the clamped-row fast K loop (NT) beside
                                 # the general one (other transpositions); 129 = one row past a tile, as the augmented matrix
                                 (129, 129, 64), (333, 257, 32)])
@pytest.mark.parametrize("pad", [True, False])
def test_gemm(env, ta, tb, mnk, pad):
    torch, hip, dev, to_dev = env
    m, n, k = mnk
    rng = np.random.default_rng(m * 1000 + n * 10 + k)
    A = rng.standard_normal((k, m) if ta else (m, k))
    B = rng.standard_normal((n, k) if tb else (k, n))
    C = rng.standard_normal((m, n))
    opA = A.T if ta else A
    opB = B.T if tb else B
    ref = 0.7 * opA @ opB - 1.3 * C
    dC = to_dev(C, pad)
    hip.gemm(to_dev(A, pad), to_dev(B, pad), ta=ta, tb=tb, alpha=0.7, beta=-1.3, out=dC)
    scale = np.abs(opA) @ np.abs(opB) + np.abs(C)
    assert np.all(np.abs(dC.cpu().numpy() - ref) <= 1e-13 * scale + 1e-300)
    # beta = 0 must not read C (NaN-filled output buffer)
    dC2 = to_dev(np.full((m, n), np.nan), pad)
    hip.gemm(to_dev(A, pad), to_dev(B, pad), ta=ta, tb=tb, alpha=1.0, beta=0.0, out=dC2)
    assert np.all(np.abs(dC2.cpu().numpy() - opA @ opB) <= 1e-13 * scale)


@pytest.mark.parametrize("m,k", [(1, 1), (5, 3), (300, 1000), (1024, 1024), (2050, 70), (7, 5000)])
def test_single_column_product_is_a_matrix_vector_pass(env, m, k, monkeypatch):
    """gemm(A, v, tb=True) with a one-row v: gpar_gemv (one wave per row); the tile path (GPAR_GEMV=0) agrees to rounding."""
    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(m + k)
    A, v = rng.standard_normal((m, k)), rng.standard_normal((1, k))
    got = hip.gemm(to_dev(A), to_dev(v), tb=True, alpha=-1.5).cpu().numpy()
    want = -1.5 * (A @ v.T)
    scale = np.abs(A) @ np.abs(v.T) + 1e-300
    assert got.shape == (m, 1) and np.max(np.abs(got - want) / scale) < 1e-14
    monkeypatch.setenv("GPAR_GEMV", "0")
    tiles = hip.gemm(to_dev(A), to_dev(v), tb=True, alpha=-1.5).cpu().numpy()
    assert np.max(np.abs(tiles - want) / scale) < 1e-14


def test_gemm_asymmetric_identity(env):
    # A = I against an asymmetric B catches transposed / permuted output maps
    torch, hip, dev, to_dev = env
    n = 200
    B = np.arange(n * n, dtype=np.float64).reshape(n, n)
    out = hip.gemm(to_dev(np.eye(n)), to_dev(B)).cpu().numpy()
    assert np.array_equal(out, B)
    out = hip.gemm(to_dev(np.eye(n)), to_dev(B), tb=True).cpu().numpy()
    assert np.array_equal(out, B.T)


@pytest.mark.parametrize("n", [70, 256, 300])
def test_gemm_lower_flags(env, n):
    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(n)
    P = rng.standard_normal((n, 40))
    C = rng.standard_normal((n, n))
    dC = to_dev(C)
    hip.gemm(to_dev(P), to_dev(P), tb=True, alpha=-1.0, beta=1.0, out=dC, c_lower=True)
    got = dC.cpu().numpy()
    ref = C - P @ P.T
    il = np.tril_indices(n)
    iu = np.triu_indices(n, 1)
    assert np.allclose(got[il], ref[il], rtol=1e-12, atol=1e-12)
    assert np.array_equal(got[iu], C[iu])  # strict upper triangle untouched
    # triangular op(A): out = tril(L) @ Z with junk above the diagonal of L
    L = rng.standard_normal((n, n))
    Z = rng.standard_normal((n, 9))
    got = hip.gemm(to_dev(L), to_dev(Z), a_lower=True).cpu().numpy()
    assert np.allclose(got, np.tril(L) @ Z, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("n", [1, 2, 17, 64, 65, 127, 128, 129, 200, 513, 1000, 1700])
def test_potrf_full(env, n):
    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(n)
    A = _spd(rng, n)
    dA = to_dev(np.tril(A) + np.triu(np.full((n, n), np.nan), 1))  # upper triangle must never be read
    logdet, info = hip.potrf_(dA)
    L = np.tril(dA.cpu().numpy())
    Lref = np.linalg.cholesky(A)
    assert int(info.item()) == 0
    assert np.allclose(L, Lref, rtol=1e-10, atol=1e-12)
    assert np.isclose(logdet.item(), 2 * np.sum(np.log(np.diag(Lref))), rtol=1e-12)


def test_potrf_not_pd_reports_info(env):
    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(0)
    n = 150
    A = _spd(rng, n)
    A[100, 100] = -5.0
    _, info = hip.potrf_(to_dev(A))
    assert int(info.item()) == 101


@pytest.mark.parametrize("N,nf", [(10, 4), (130, 64), (300, 257), (700, 300), (1400, 1399)])
def test_potrf_partial_schur(env, N, nf):
    """Factor the leading nf columns; the trailing block must hold the Schur complement (this is what turns
    one routine into logpdf + posterior mean + posterior covariance)."""
    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(N + nf)
    A = _spd(rng, N)
    dA = to_dev(A)
    hip.potrf_(dA, nf=nf)
    got = dA.cpu().numpy()
    L11 = np.linalg.cholesky(A[:nf, :nf])
    L21 = scipy.linalg.solve_triangular(L11, A[:nf, nf:], lower=True).T
    S = A[nf:, nf:] - L21 @ L21.T
    assert np.allclose(np.tril(got[:nf, :nf]), L11, rtol=1e-10, atol=1e-12)
    assert np.allclose(got[nf:, :nf], L21, rtol=1e-9, atol=1e-11)
    il = np.tril_indices(N - nf)
    assert np.allclose(got[nf:, nf:][il], S[il], rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("n,rows", [(1, 1), (50, 3), (64, 64), (200, 1), (333, 130), (1000, 70), (1024, 300), (1536, 130), (2100, 700), (1100, 64)])
def test_trsm(env, n, rows):
    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(n + rows)
    L = np.linalg.cholesky(_spd(rng, n))
    B = rng.standard_normal((rows, n))
    dL = to_dev(L + np.triu(np.full((n, n), np.nan), 1))
    X = hip.trsm_rlt_(dL, to_dev(B)).cpu().numpy()
    assert np.allclose(X, scipy.linalg.solve_triangular(L, B.T, lower=True).T, rtol=1e-9, atol=1e-11)
    # (from n = 1024 and 64 rows on: the fused backward block kernel + one K = 512 update per block; 2100 has a ragged tail)
    X = hip.trsm_rln_(dL, to_dev(B)).cpu().numpy()
    assert np.allclose(X, scipy.linalg.solve_triangular(L, B.T, lower=True, trans="T").T, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("n,rows", [(50, 3), (333, 130), (1024, 300), (1100, 64), (840, 841), (5000, 70)])
def test_predicated_solve_and_pivot_spread(env, n, rows):
    """gpar_trsm_rlt_if: the solve happens iff (flag != 0) == run_if, decided on the device - either exactly the unconditional
    solve (same bits) or nothing at all (B untouched), through every kernel the path launches (fused blocks, their GEMM updates,
    ragged strips, paired blocks).  gpar_chol_spread: (max L_jj / min L_jj)^2 and the flag against a limit."""
    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(n * 7 + rows)
    L = np.linalg.cholesky(_spd(rng, n))
    B = rng.standard_normal((rows, n))
    dL = to_dev(L + np.triu(np.full((n, n), np.nan), 1))
    want = hip.trsm_rlt_(dL, to_dev(B)).cpu().numpy()
    diag = np.diag(L)
    true_spread = (diag.max() / diag.min()) ** 2
    for limit, expect in ((true_spread * 1.0001, 0), (true_spread * 0.9999, 1)):
        spread, flag = hip.chol_spread(dL, limit)
        assert int(flag.item()) == expect and abs(float(spread) - true_spread) <= 1e-12 * true_spread
        for sense in (False, True):
            X = hip.trsm_rlt_(dL, to_dev(B), when=(flag, sense)).cpu().numpy()
            assert np.array_equal(X, want if bool(expect) == sense else B), (limit, sense)
    bad = to_dev(L + np.triu(np.full((n, n), np.nan), 1))
    bad[n // 2, n // 2] = float("nan")   # a failed factorisation counts as ill-conditioned
    assert int(hip.chol_spread(bad, 1e300)[1].item()) == 1


@pytest.mark.parametrize("n,rows", [(5000, 130), (6145, 70)])
def test_trsm_forward_with_paired_blocks(env, n, rows):
    """n >= 4096 + 1024: the forward solve pairs 512-column blocks (one rank-1024 update per pair), also with a ragged
    tail; the triangular-aware inverse built on it is checked against numpy at the same size."""
    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(n)
    L = np.linalg.cholesky(_spd(rng, n, cond=50.0))
    B = rng.standard_normal((rows, n))
    dL = to_dev(L + np.triu(np.full((n, n), np.nan), 1))
    X = hip.trsm_rlt_(dL, to_dev(B)).cpu().numpy()
    assert np.allclose(X, scipy.linalg.solve_triangular(L, B.T, lower=True).T, rtol=1e-9, atol=1e-10)
    Kinv = np.tril(hip.chol_inverse(dL).cpu().numpy())
    ref = np.tril(np.linalg.inv(L @ L.T))
    assert np.allclose(Kinv, ref, rtol=1e-8, atol=1e-9 * np.abs(ref).max())


@pytest.mark.parametrize("n", [128, 130, 511, 512, 513, 1090, 1700])
def test_single_row_backward_solve_takes_the_trsv_path(env, n):
    """alpha^T = z^T L^-1 for one row (gpar_trsm_rln with nrows = 1, n >= 128) runs the dedicated TRSV kernels:
    ragged last block, ragged 64-column sub-block, poisoned upper triangle."""
    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(n)
    L = np.linalg.cholesky(_spd(rng, n))
    b = rng.standard_normal((1, n))
    dL = to_dev(L + np.triu(np.full((n, n), np.nan), 1))
    x = hip.trsm_rln_(dL, to_dev(b)).cpu().numpy()
    assert np.allclose(x, scipy.linalg.solve_triangular(L, b.T, lower=True, trans="T").T, rtol=1e-9, atol=1e-11)


def test_randn_matches_philox_oracle(env):
    torch, hip, dev, to_dev = env
    from oracle import philox

    for rows, cols, seed, off in [(1, 1, 1, 0), (7, 3, 42, 5), (100, 33, 2**40 + 3, 2**33)]:
        got = hip.randn(seed, off, rows, cols, dev).cpu().numpy()
        ref = philox.randn(seed, off, rows, cols)
        assert np.allclose(got, ref, rtol=0, atol=1e-13)
    z = hip.randn(3, 0, 2000, 500, dev).cpu().numpy()
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3


@pytest.mark.parametrize("S,shape", [(2, (5, 3)), (7, (1, 1)), (30, (40, 8)), (100, (64, 5)), (257, (300, 2))])
def test_sample_stats_is_numpy_mean_and_percentile(env, S, shape):
    """gpar_sample_stats vs np.mean / np.percentile (reference regression.py:589-595): bit-exact, ties included."""
    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(S)
    x = rng.standard_normal((S,) + shape)
    x[S // 2] = x[0]  # exact ties between samples
    x[:, 0, 0] = 1.25  # a constant column
    d = torch.tensor(x, dtype=torch.float64, device=dev)
    mean, lo, hi = hip.sample_stats(d, 2.5, 97.5)
    assert np.array_equal(mean.cpu().numpy(), np.mean(x, axis=0))
    assert np.array_equal(lo.cpu().numpy(), np.percentile(x, 2.5, axis=0))
    assert np.array_equal(hi.cpu().numpy(), np.percentile(x, 97.5, axis=0))
    for q_lo, q_hi in [(0.0, 100.0), (50.0, 50.0), (33.3, 66.6)]:
        _, lo, hi = hip.sample_stats(d, q_lo, q_hi)
        assert np.array_equal(lo.cpu().numpy(), np.percentile(x, q_lo, axis=0))
        assert np.array_equal(hi.cpu().numpy(), np.percentile(x, q_hi, axis=0))
    only_mean = hip.sample_stats(d)
    assert only_mean[1] is None and only_mean[2] is None and np.array_equal(only_mean[0].cpu().numpy(), np.mean(x, axis=0))


@pytest.mark.parametrize("n", [1, 63, 64, 65, 300, 2048])
def test_trmv_lower(env, n):
    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(n)
    L = np.tril(rng.standard_normal((n, n)))
    x = rng.standard_normal((n, 3))
    dL = to_dev(L + np.triu(np.full((n, n), np.nan), 1))  # the strict upper triangle must not be read
    dx = to_dev(x)
    got = hip.trmv_lower(dL, dx[:, 1:2]).cpu().numpy()  # a strided column
    np.testing.assert_allclose(got, L @ x[:, 1:2], rtol=1e-12, atol=1e-12 * np.sqrt(n))


@pytest.mark.parametrize("n", [1, 63, 64, 65, 129, 300, 2049])
def test_trmv_upper_and_the_alpha_it_stands_for(env, n):
    """gpar_trmv_upper (ABI v7): U x for an upper-triangular U whose strict lower triangle is never read; with U = the workspace
    gpar_chol_inverse leaves (L^-T) and x = L^-1 y it is (L L^T)^-1 y - the alpha of the gradient's weights - to the accuracy of the
    backward substitution it replaces."""
    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(n)
    U = np.triu(rng.standard_normal((n, n)))
    x = rng.standard_normal((n, 3))
    dU = to_dev(U + np.tril(np.full((n, n), np.nan), -1))
    got = hip.trmv_upper(dU, to_dev(x)[:, 1]).cpu().numpy()   # a strided vector
    np.testing.assert_allclose(got, U @ x[:, 1], rtol=1e-12, atol=1e-12 * np.sqrt(n))
    G = rng.standard_normal((n, n + 5))
    K = G @ G.T / n + 0.5 * np.eye(n)
    y = rng.standard_normal(n)
    Ld = to_dev(K)
    _, info = hip.potrf_(Ld)
    assert int(info.item()) == 0
    Kinv, X = hip.chol_inverse(Ld, with_x=True)
    L = np.linalg.cholesky(K)
    z = np.linalg.solve(L, y)
    alpha = hip.trmv_upper(X, to_dev(z)).cpu().numpy()
    want = np.linalg.solve(K, y)
    np.testing.assert_allclose(alpha, want, rtol=1e-10, atol=1e-11 * np.abs(want).max())
    np.testing.assert_allclose(np.tril(Kinv.cpu().numpy()), np.tril(np.linalg.inv(K)), rtol=1e-9, atol=1e-11)


def test_reductions(env):
    """gpar_dot, gpar_gemv_t (A^T v for a tall A), gpar_rownorm2: ragged sizes, padded and unpadded storage."""
    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(5)
    x, y = rng.standard_normal(5000), rng.standard_normal(5000)
    d = hip.dot(to_dev(x), 1, to_dev(y), 1, 5000)
    assert np.isclose(d.item(), x @ y, rtol=1e-12)
    for rows, cols, pad in [(1, 1, True), (7, 3, True), (129, 513, True), (1000, 37, False), (2049, 1024, True), (300, 1025, False), (5000, 130, True)]:
        A = rng.standard_normal((rows, cols))
        v = rng.standard_normal(rows)
        scale = np.abs(A).T @ np.abs(v)
        got = hip.gemv_t(to_dev(A, pad), to_dev(v)).cpu().numpy()
        assert np.all(np.abs(got - A.T @ v) <= 1e-13 * scale + 1e-300), (rows, cols)
        again = hip.gemv_t(to_dev(A, pad), to_dev(v)).cpu().numpy()
        assert np.array_equal(got, again)  # fixed reduction order
        got = hip.rownorm2(to_dev(A, pad)).cpu().numpy()
        np.testing.assert_allclose(got, np.sum(A * A, axis=1), rtol=1e-13)
    from gpar_amd import _lib

    lib = _lib.load()
    assert lib.gpar_workspace_doubles(_lib.WS_GEMM_SPLITK, 10, 20, 3) == 600
    assert lib.gpar_workspace_doubles(_lib.WS_GEMV_T, 513, 7, 0) == 2 * 7
    assert lib.gpar_workspace_doubles(_lib.WS_GRAM_GRAD, 5, 0, 0) == 5 * _lib.GRAD_NACC
    assert lib.gpar_workspace_doubles(99, 1, 1, 1) == -1


def test_sample_stats_propagates_nan_like_numpy(env):
    """A NaN among an element's samples: np.percentile gives NaN bounds (rank counting mis-ranked instead)."""
    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(11)
    x = rng.standard_normal((20, 6, 2))
    x[3, 2, 1] = np.nan
    x[7, 4, 0] = np.inf
    d = torch.tensor(x, dtype=torch.float64, device=dev)
    mean, lo, hi = hip.sample_stats(d, 2.5, 97.5)
    with np.errstate(invalid="ignore"):
        np.testing.assert_array_equal(mean.cpu().numpy(), np.mean(x, axis=0))
        np.testing.assert_array_equal(lo.cpu().numpy(), np.percentile(x, 2.5, axis=0))
        np.testing.assert_array_equal(hi.cpu().numpy(), np.percentile(x, 97.5, axis=0))
    assert np.isnan(lo.cpu().numpy()[2, 1]) and np.isnan(hi.cpu().numpy()[2, 1])


def _kernels(m, p_cols):
    """A zoo of GPAR-shaped kernels over m inputs and the given output columns."""
    from gpar_amd.kernels import EQ, RQ, Linear, ZeroKernel

    rng = np.random.default_rng(m + len(p_cols))
    mi = list(range(m))
    s = lambda k: rng.uniform(0.5, 2.0, k)
    zoo = {}
    zoo["eq"] = (1.3 * EQ().stretch(s(m))).select(mi)
    zoo["rq"] = (0.7 * RQ(0.4).stretch(s(m))).select(mi)
    zoo["eq+lin+const"] = (2.0 * EQ().stretch(s(m)) + Linear().stretch(s(m)) + 0.5).select(mi)
    zoo["locally-periodic"] = (
        1.1 * EQ().stretch(s(m)) + 0.9 * EQ().stretch(s(2 * m)).periodic(s(m)) * EQ().stretch(10 * s(m))
    ).select(mi)
    if p_cols:
        k_out = Linear().stretch(s(len(p_cols))) + 0.8 * EQ().stretch(s(len(p_cols)))
        zoo["gpar-layer"] = (1.0 * EQ().stretch(s(m))).select(mi) + k_out.select(p_cols)
        zoo["gpar-layer-rq"] = (1.0 * RQ(1.5).stretch(s(m))).select(mi) + (
            Linear().stretch(s(len(p_cols))) + 0.8 * RQ(0.2).stretch(s(len(p_cols)))
        ).select(p_cols)
    # markov=0 quirk: output kernels over zero columns: EQ -> 1, Linear -> 0
    zoo["zero-width-outputs"] = (1.0 * EQ().stretch(s(m))).select(mi) + (
        Linear().stretch(np.zeros(0)) + 0.6 * EQ().stretch(np.zeros(0))
    ).select([])
    zoo["zero"] = ZeroKernel()
    return zoo


@pytest.mark.parametrize("m,p_cols", [(1, []), (2, [2]), (3, [3, 4, 5]), (4, [6, 7])])
@pytest.mark.parametrize("n1,n2", [(1, 1), (50, 70), (64, 64), (129, 200)])
def test_gram_matches_oracle(env, m, p_cols, n1, n2):
    torch, hip, dev, to_dev = env
    from gpar_amd.kernels import compile_kernel
    from oracle import kernels as ok

    width = m + (max(p_cols) - m + 1 if p_cols else 0)
    rng = np.random.default_rng(n1 * 7 + n2)
    x1, x2 = rng.standard_normal((n1, width)), rng.standard_normal((n2, width))
    for name, k in _kernels(m, p_cols).items():
        ck = compile_kernel(k, width)
        spec = ok.spec_to_dict(k.resolve(width))
        z1, z2 = hip.featurize(ck, to_dev(x1)), hip.featurize(ck, to_dev(x2))
        got = hip.gram(ck, z1, z2).cpu().numpy()
        ref = ok.gram(spec, x1, x2)
        assert np.allclose(got, ref, rtol=1e-13, atol=1e-14), name
        # rows scaled in the same pass (the D^-1/2 K_xz of the inducing-point path)
        rs = rng.uniform(0.5, 3.0, n1)
        got = hip.gram(ck, z1, z2, row_scale=to_dev(rs)).cpu().numpy()
        assert np.allclose(got, ref * rs[:, None], rtol=1e-13, atol=1e-14), name
        # symmetric, lower-only, with noise diagonal + jitter
        noise = rng.uniform(0.01, 0.1, n1)
        K = to_dev(np.full((n1, n1), np.nan))
        hip.gram(ck, z1, None, out=K, lower=True, diag_add=to_dev(noise), diag_const=1e-12)
        got = K.cpu().numpy()
        ref = ok.gram(spec, x1, None, noise_diag=noise, jitter=1e-12)
        il = np.tril_indices(n1)
        assert np.allclose(got[il], ref[il], rtol=1e-13, atol=1e-14), name
        assert np.allclose(hip.gram_diag(ck, z1).cpu().numpy(), ok.gram_diag(spec, x1), rtol=1e-13, atol=1e-14), name


@pytest.mark.parametrize("m,p_cols", [(1, []), (2, [2]), (3, [3, 4, 5])])
@pytest.mark.parametrize("n", [5, 64, 130, 300])
def test_kernel_gradients_match_oracle(env, m, p_cols, n):
    """Fused device gradient pass (moment sums + host chain rule) vs the oracle's explicit dK/dtheta matrices."""
    torch, hip, dev, to_dev = env
    from gpar_amd.engine import HipEngine
    from gpar_amd.kernels import compile_kernel
    from oracle import kernels as ok

    eng = HipEngine()
    width = m + (max(p_cols) - m + 1 if p_cols else 0)
    rng = np.random.default_rng(n + m)
    x = rng.standard_normal((n, width))
    Wfull = rng.standard_normal((n, n))
    Wfull = Wfull + Wfull.T
    Wdev = to_dev(np.tril(Wfull) + np.triu(np.full((n, n), np.nan), 1))
    for name, k in _kernels(m, p_cols).items():
        if name == "zero":
            continue
        ck = compile_kernel(k, width)
        got = eng.kernel_grads(ck, to_dev(x), Wdev)
        ref = ok.kernel_grads(ok.spec_to_dict(k.resolve(width)), x, Wfull)
        scale = np.abs(Wfull).sum()
        for t in range(len(ref["coef"])):
            assert abs(got["coef"][t] - ref["coef"][t]) <= 1e-12 * scale, (name, "coef", t)
            for fi, (gf, rf) in enumerate(zip(got["factors"][t], ref["factors"][t])):
                for key in ("scales", "periods", "alpha"):
                    if rf[key] is None:
                        continue
                    assert np.allclose(gf[key], rf[key], rtol=1e-10, atol=1e-12 * scale), (name, key, t, fi)


@pytest.mark.parametrize("n,M", [(70, 9), (257, 65), (130, 200)])
def test_cross_and_diagonal_gradient_passes_match_oracle(env, n, M):
    """gpar_gram_grad_cross in its three modes (rectangular n x M weights, symmetric M x M, diagonal) through
    HipEngine.kernel_grads_vfe vs the oracle's explicit derivative matrices on the stacked points: ragged tiles,
    more columns than rows, every kernel family."""
    torch, hip, dev, to_dev = env
    from gpar_amd.engine import HipEngine
    from gpar_amd.kernels import compile_kernel
    from oracle.engine import OracleEngine

    eng, ora = HipEngine(), OracleEngine()
    m, p_cols = 2, [2, 3]
    width = 4
    rng = np.random.default_rng(n + M)
    x, z = rng.standard_normal((n, width)), rng.standard_normal((M, width))
    Wfu = rng.standard_normal((n, M))
    Wuu = rng.standard_normal((M, M))
    Wuu = Wuu + Wuu.T
    wd = rng.standard_normal(n)
    for name, k in _kernels(m, p_cols).items():
        if name == "zero":
            continue
        ck = compile_kernel(k, width)
        got = eng.kernel_grads_vfe(ck, to_dev(x), to_dev(z), to_dev(Wfu), to_dev(np.tril(Wuu) + np.triu(np.full((M, M), np.nan), 1)),
                                   torch.tensor(wd, device=dev))
        ref = ora.kernel_grads_vfe(ora.compile(k, width), torch.tensor(x), torch.tensor(z), torch.tensor(Wfu), torch.tensor(Wuu), torch.tensor(wd))
        scale = np.abs(Wfu).sum() + np.abs(Wuu).sum() + np.abs(wd).sum()
        for t in range(len(ref["coef"])):
            assert abs(got["coef"][t] - ref["coef"][t]) <= 1e-12 * scale, (name, "coef", t)
            for fi, (gf, rf) in enumerate(zip(got["factors"][t], ref["factors"][t])):
                for key in ("scales", "periods", "alpha"):
                    if rf[key] is None:
                        continue
                    assert np.allclose(gf[key], rf[key], rtol=1e-10, atol=1e-12 * scale), (name, key, t, fi)


@pytest.mark.parametrize("n", [1, 50, 64, 129, 700, 1500, 4200, 1024, 1536, 2560, 4096])
def test_chol_inverse(env, n):
    """Triangular-aware (L L^T)^-1: L^-T by the two-level TRSM of the identity - or, for multiples of 512, by recursive blocked
    inversion (batched diagonal blocks, then levels of batched triangular-aware products; 1536 and 2560 have short last
    blocks) - + SYRK that starts k at the tile's first row."""
    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(n)
    A = _spd(rng, n, cond=50.0)
    L = np.linalg.cholesky(A)
    got = hip.chol_inverse(to_dev(L + np.triu(np.full((n, n), np.nan), 1))).cpu().numpy()
    ref = np.linalg.inv(A)
    il = np.tril_indices(n)
    assert np.allclose(got[il], ref[il], rtol=1e-9, atol=1e-11 * np.abs(ref).max())


@pytest.mark.parametrize("n", [1, 77, 640, 1500])
def test_fused_dense_logpdf_entry_point(env, n):
    """gpar_logpdf_dense: features, Gram + noise + jitter, observations, partial factorisation and value in one call equal the
    numpy / scipy evaluation of log N(y; 0, K + D + eps I); the buffer it leaves holds the factor and L^-1 y."""
    import scipy.linalg as sla

    from gpar_amd.kernels import EQ, Linear, compile_kernel

    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(n)
    x = rng.uniform(0, 1, (n, 3))
    y = rng.standard_normal(n)
    noise = rng.uniform(0.05, 0.2, n)
    kernel = 1.3 * EQ().stretch(np.array([0.4, 0.7])).select([0, 1]) + Linear().stretch(np.array([2.0])).select([2])
    ck = compile_kernel(kernel, 3)
    value, logdet, info, A = hip.logpdf_dense(ck, to_dev(x), to_dev(y[:, None])[:, 0], to_dev(noise[:, None])[:, 0], 1e-10)
    assert int(info.item()) == 0
    d2 = ((x[:, None, :2] - x[None, :, :2]) ** 2 / np.array([0.4, 0.7]) ** 2).sum(-1)
    K = 1.3 * np.exp(-0.5 * d2) + np.outer(x[:, 2], x[:, 2]) / 4.0 + np.diag(noise) + 1e-10 * np.eye(n)
    L = np.linalg.cholesky(K)
    zr = sla.solve_triangular(L, y, lower=True)
    ref = -0.5 * (2 * np.log(np.diag(L)).sum() + n * np.log(2 * np.pi) + zr @ zr)
    assert abs(float(value) - ref) <= 1e-10 * max(1.0, abs(ref))
    got = A.cpu().numpy()
    assert np.allclose(np.tril(got[:n, :n]), L, rtol=1e-9, atol=1e-11)
    assert np.allclose(got[n, :n], zr, rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("N,nf,batch", [(1024, 1024, 3), (1601, 1600, 4), (700, 700, 2), (1100, 1038, 3), (100, 100, 2), (2048, 1024, 5)])
def test_lockstep_batch_factorisation_equals_one_at_a_time(env, N, nf, batch):
    """gpar_potrf_batch: `batch` (partial) factorisations in lock-step - one panel launch and one batched trailing update per 512
    columns - leave in every matrix what gpar_potrf leaves in it alone (factor, Schur complement, logdet, info), and one
    indefinite matrix in the batch is reported in its own info word without disturbing the others."""
    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(N + batch)
    mats = []
    for b in range(batch):
        G = rng.standard_normal((N, N + 8))
        mats.append(G @ G.T / N + (0.5 + b) * np.eye(N))
    stacked = hip.alloc_matrix(batch * N, N, dev)
    stacked.copy_(to_dev(np.concatenate(mats, axis=0)))
    logdet, info = hip.potrf_batch_(stacked, batch, nf)
    assert info.cpu().tolist() == [0] * batch
    got = stacked.cpu().numpy()
    for b in range(batch):
        single = to_dev(mats[b]).clone()
        ld1, info1 = hip.potrf_(single, nf)
        assert int(info1.item()) == 0
        ref = single.cpu().numpy()
        blk = got[b * N:(b + 1) * N]
        assert np.allclose(np.tril(blk)[:, :nf], np.tril(ref)[:, :nf], rtol=1e-12, atol=1e-13)
        assert np.allclose(np.tril(blk[nf:, nf:]), np.tril(ref[nf:, nf:]), rtol=1e-10, atol=1e-12)
        assert abs(float(logdet[b]) - float(ld1)) <= 1e-11 * max(1.0, abs(float(ld1)))
    # one indefinite member
    bad = [m.copy() for m in mats]
    j = min(nf - 1, 70)
    bad[1][j, j] = -1.0
    stacked.copy_(to_dev(np.concatenate(bad, axis=0)))
    logdet, info = hip.potrf_batch_(stacked, batch, nf)
    flags = info.cpu().tolist()
    assert flags[1] == j + 1 and all(v == 0 for i, v in enumerate(flags) if i != 1)
    got = stacked.cpu().numpy()
    L0 = np.linalg.cholesky(mats[0])
    assert np.allclose(np.tril(got[:N])[:, :nf], L0[:, :nf], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("N,nf,batch", [(1025, 1024, 1), (1026, 1024, 4), (2049, 2048, 1), (2100, 2048, 3), (3137, 3136, 1), (2600, 1100, 2), (5200, 5200, 1),
                                        (4161, 4160, 2)])
def test_several_panels_per_launch_leave_the_factor_of_one_panel_per_launch(env, monkeypatch, N, nf, batch):
    """potrf_group_kernel (two to GPAR_POTRF_FUSE_MAX 512-column panels per launch, each panel's chain starting one tile product
    behind the one before; GPAR_POTRF_FUSE2_ROWS / GPAR_POTRF_FUSE2_BATCH_ROWS) against the one-panel-per-launch schedule: same
    factor, Schur complement and logdet to rounding, with EVERY eligible step fused in twos, threes and eights and with the default rule; single augmented rows (N = 64 k + 1),
    ragged last row blocks, partial factorisations with many rows below, lock-step batches, look-ahead on (N = 5200); repeated
    runs return the same bits (the hand-off words are never reset inside a factorisation)."""
    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(N + batch)
    pts = rng.uniform(0, 1, (N, 3))
    mats = []
    for b in range(batch):
        d2 = ((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
        mats.append(np.exp(-0.5 * d2 / (0.2 + 0.05 * b)) + (0.05 + 0.02 * b) * np.eye(N))
    host = to_dev(np.concatenate(mats, axis=0))
    A = hip.alloc_matrix(batch * N, N, dev)

    def run():
        A.copy_(host)
        if batch == 1:
            logdet, info = hip.potrf_(A, nf)
        else:
            logdet, info = hip.potrf_batch_(A, batch, nf)
        assert info.cpu().tolist() == [0] * batch
        return np.tril(A.cpu().numpy().reshape(batch, N, -1)[:, :, :N]), logdet.cpu().numpy().copy()

    monkeypatch.setenv("GPAR_POTRF_FUSE2_ROWS", "0")
    monkeypatch.setenv("GPAR_POTRF_FUSE2_BATCH_ROWS", "0")
    ref, ref_logdet = run()
    for rows, most in (("1000000", "2"), ("1000000", "3"), ("1000000", "8"), (None, None)):
        for name, value in (("GPAR_POTRF_FUSE2_ROWS", rows), ("GPAR_POTRF_FUSE2_BATCH_ROWS", rows), ("GPAR_POTRF_FUSE_MAX", most)):
            if value is None:
                monkeypatch.delenv(name, raising=False)
            else:
                monkeypatch.setenv(name, value)
        got, logdet = run()
        scale = np.abs(ref).max()
        assert np.abs(got - ref).max() <= 2e-13 * scale, (rows, np.abs(got - ref).max() / scale)
        assert np.allclose(logdet, ref_logdet, rtol=1e-13, atol=0)
        again, logdet2 = run()
        assert np.array_equal(again, got) and np.array_equal(logdet2, logdet)


@pytest.mark.parametrize("N,nf", [(2601, 2600), (3137, 3136), (4097, 4096), (5001, 5000), (5200, 5200), (6300, 6200), (4700, 3000)])
def test_look_ahead_does_not_change_a_single_bit_of_the_factor(env, N, nf):
    """The schedule of a factorisation is a function of its shape only: with the trailing update on the side stream (look-ahead) or
    on the caller's, the same launches with the same tile shapes touch every element in the same order - sizes at which single
    panels, fused launches of several panels, ragged last panels and the small update kernels all occur."""
    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(N)
    pts = to_dev(rng.uniform(0, 1, (N, 3)))
    K = hip.alloc_matrix(N, N, dev)
    K.copy_(torch.exp(-0.5 * torch.cdist(pts, pts) ** 2 / 0.2))
    K.diagonal().add_(0.05)
    A = hip.alloc_matrix(N, N, dev)
    out = []
    for la in (True, False, True):
        A.copy_(K)
        logdet, info = hip.potrf_(A, nf, lookahead=la)
        assert int(info.item()) == 0
        out.append((torch.tril(A).clone(), float(logdet)))
    assert torch.equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]
    assert torch.equal(out[0][0], out[2][0]) and out[0][1] == out[2][1]


@pytest.mark.parametrize("ns,n,batch", [(200, 333, 5), (512, 1024, 3), (65, 10, 2)])
def test_batched_downdate_and_draws(env, ns, n, batch):
    """gpar_gemm_batch (C_b -= V_b V_b^T, lower) and gpar_trmv_lower_batch (y_b = L_b z_b + m_b): the per-sample steps of
    ancestral sampling for all samples of a layer in one launch each, equal to the per-sample calls."""
    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(ns + n)
    V = rng.standard_normal((batch * ns, n)) / np.sqrt(n)
    C = rng.standard_normal((batch * ns, ns))
    dV, dC = to_dev(V), to_dev(C)
    ref = []
    for b in range(batch):
        blk = dC[b * ns:(b + 1) * ns].clone()
        hip.gemm(dV[b * ns:(b + 1) * ns], dV[b * ns:(b + 1) * ns], tb=True, alpha=-1.0, beta=1.0, out=blk, c_lower=True)
        ref.append(np.tril(blk.cpu().numpy()))
    hip.gemm_batch_(dV, dV, dC, batch, tb=True, alpha=-1.0, beta=1.0, c_lower=True)
    got = dC.cpu().numpy()
    for b in range(batch):
        assert np.array_equal(np.tril(got[b * ns:(b + 1) * ns]), ref[b])
        assert np.allclose(ref[b], np.tril(C[b * ns:(b + 1) * ns] - V[b * ns:(b + 1) * ns] @ V[b * ns:(b + 1) * ns].T), rtol=1e-12, atol=1e-12)
    # gpar_gram_batch: the prior covariances of all samples in one launch
    from gpar_amd.kernels import EQ, Linear, compile_kernel

    ck = compile_kernel(0.7 * EQ().stretch(np.array([0.5, 0.8])).select([0, 1]) + Linear().stretch(np.array([3.0])).select([2]), 3)
    xs = to_dev(rng.uniform(0, 1, (batch * ns, 3)))
    z_all = hip.featurize(ck, xs)
    noise = to_dev(rng.uniform(0.1, 0.2, (ns, 1)))[:, 0]
    Ks = hip.alloc_matrix(batch * ns, ns, dev)
    hip.gram_batch_(ck, z_all, batch, Ks, lower=True, diag_add=noise, diag_const=1e-9)
    for b in range(batch):
        one = hip.gram(ck, z_all[b * ns:(b + 1) * ns], lower=True, diag_add=noise, diag_const=1e-9)
        assert torch.equal(torch.tril(Ks[b * ns:(b + 1) * ns]), torch.tril(one))
    Ls = np.tril(rng.standard_normal((batch * ns, ns)))
    Z = rng.standard_normal((ns, batch))
    M = rng.standard_normal((batch * ns, 1))
    dL, dZ, dM = to_dev(Ls), to_dev(Z), to_dev(M)
    out = torch.empty(ns, batch, dtype=torch.float64, device=dev)
    hip.trmv_lower_batch_(dL, batch, dZ, out, add=dM)
    plain = torch.empty(ns, batch, dtype=torch.float64, device=dev)
    hip.trmv_lower_batch_(dL, batch, dZ, plain)
    for b in range(batch):
        single = hip.trmv_lower(dL[b * ns:(b + 1) * ns], dZ[:, b:b + 1])
        assert torch.equal(plain[:, b:b + 1], single)
        assert torch.equal(out[:, b:b + 1], single + dM[b * ns:(b + 1) * ns])
        assert np.allclose(single.cpu().numpy()[:, 0], np.tril(Ls[b * ns:(b + 1) * ns]) @ Z[:, b], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("n,batch", [(640, 3), (1500, 4), (77, 2)])
def test_lockstep_dense_logpdf_equals_layer_by_layer(env, n, batch):
    """hip.logpdf_dense_batch (build per layer, one lock-step factorisation, one finishing launch) returns per layer what
    gpar_logpdf_dense returns for it alone; the layers have different kernels and input widths, as GPAR's layers do."""
    from gpar_amd.kernels import EQ, Linear, compile_kernel

    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(n)
    items, ref = [], []
    for b in range(batch):
        width = 2 + b
        x = to_dev(rng.uniform(0, 1, (n, width)))
        y = to_dev(rng.standard_normal((n, 1)))[:, 0]
        noise = to_dev(rng.uniform(0.05, 0.2, (n, 1)))[:, 0] if b % 2 == 0 else None
        kernel = (1.0 + 0.3 * b) * EQ().stretch(np.full(width - 1, 0.5)).select(list(range(width - 1))) + \
            Linear().stretch(np.array([2.0])).select([width - 1])
        ck = compile_kernel(kernel, width)
        items.append((ck, x, y, noise))
        value, _, info, _ = hip.logpdf_dense(ck, x, y, noise, 1e-8 if noise is not None else 1e-2)
        assert int(info.item()) == 0
        ref.append(float(value))
    # the jitter is one number per call: evaluate the two groups of layers that share it
    for group, jitter in (([b for b in range(batch) if b % 2 == 0], 1e-8), ([b for b in range(batch) if b % 2 == 1], 1e-2)):
        if not group:
            continue
        vals, info = hip.logpdf_dense_batch([items[b] for b in group], jitter)
        assert info.cpu().tolist() == [0] * len(group)
        for v, b in zip(vals.cpu().tolist(), group):
            assert abs(v - ref[b]) <= 1e-11 * max(1.0, abs(ref[b]))


@pytest.mark.parametrize("n", [384, 1024, 1300])
def test_gemm_triangular_aware_k_ranges(env, n):
    """K_FROM_ROW (upper-triangular op(A), zeros stored left of its diagonal) and K_TO_COL (upper-triangular op(B), zeros stored
    below its diagonal) only shorten the K loop: same numbers as the plain product; the two factors of the recursive
    triangular inversion (gpar_chol_inverse) are this pair.  (1024: tile counts that take the per-XCD column grouping.)"""
    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(n)
    U = np.triu(rng.standard_normal((n, n)))
    D = rng.standard_normal((n, n))
    dU, dD = to_dev(U), to_dev(D)
    got = hip.gemm(dU, dD, tb=True, k_from_row=True).cpu().numpy()       # U D^T
    assert np.allclose(got, U @ D.T, rtol=1e-11, atol=1e-11 * n)
    got = hip.gemm(dD, dU, k_to_col=True, alpha=-1.0).cpu().numpy()       # -D U
    assert np.allclose(got, -(D @ U), rtol=1e-11, atol=1e-11 * n)


@pytest.mark.parametrize("ta,tb", [(True, False), (False, False), (False, True)])
@pytest.mark.parametrize("mnk", [(1, 300, 9000), (257, 130, 20000), (1025, 1025, 16411)])
def test_gemm_split_k_path(env, ta, tb, mnk):
    """Few output tiles + very long K takes the split-K path (partial slabs + ordered reduce): same answer, and
    bit-for-bit repeatable."""
    torch, hip, dev, to_dev = env
    m, n, k = mnk
    rng = np.random.default_rng(m + n + k)
    A = rng.standard_normal((k, m) if ta else (m, k))
    B = rng.standard_normal((n, k) if tb else (k, n))
    opA, opB = (A.T if ta else A), (B.T if tb else B)
    dA, dB = to_dev(A), to_dev(B)
    got = hip.gemm(dA, dB, ta=ta, tb=tb).cpu().numpy()
    again = hip.gemm(dA, dB, ta=ta, tb=tb).cpu().numpy()
    assert np.array_equal(got, again)
    scale = np.abs(opA) @ np.abs(opB)
    assert np.all(np.abs(got - opA @ opB) <= 2e-13 * scale)
    if m == n:
        C = rng.standard_normal((m, n))
        dC = to_dev(C)
        hip.gemm(dA, dB, ta=ta, tb=tb, alpha=0.5, beta=2.0, out=dC, c_lower=True)
        il, iu = np.tril_indices(m), np.triu_indices(m, 1)
        gotc = dC.cpu().numpy()
        assert np.all(np.abs(gotc - (0.5 * opA @ opB + 2 * C))[il] <= 2e-13 * (scale + np.abs(C))[il])
        assert np.array_equal(gotc[iu], C[iu])


@pytest.mark.parametrize("n", [1, 5, 64, 257, 1000])
def test_pack_and_unpack_lower(env, n):
    torch, hip, dev, to_dev = env
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n))
    dA = to_dev(A)
    packed = hip.pack_lower(dA).cpu().numpy()
    assert np.array_equal(packed, A[np.tril_indices(n)])
    B = to_dev(np.full((n, n), 7.0))
    hip.unpack_lower_(to_dev(packed), B)
    got = B.cpu().numpy()
    assert np.array_equal(np.tril(got), np.tril(A))
    assert np.all(got[np.triu_indices(n, 1)] == 7.0)  # the strict upper triangle is left alone


@pytest.mark.parametrize("n", [700, 4096])
def test_unfused_factorisation_path_agrees(env, n):
    """gpar_potrf_ex(GPAR_POTRF_UNFUSED): the retry path after a hand-off timeout (separate leaf kernels) against the
    default path."""
    torch, hip, dev, to_dev = env
    g = torch.Generator().manual_seed(n)
    x = torch.rand(n, 3, generator=g, dtype=torch.float64).to(dev)
    K = torch.exp(-0.5 * torch.cdist(x, x) ** 2 / 0.25)
    K.diagonal().add_(0.1)
    A, B = hip.alloc_matrix(n, n, dev), hip.alloc_matrix(n, n, dev)
    A.copy_(K)
    B.copy_(K)
    la, ia = hip.potrf_(A)
    lb, ib = hip.potrf_(B, fused=False, lookahead=False)
    assert int(ia.item()) == 0 and int(ib.item()) == 0
    assert abs(float(la) - float(lb)) <= 1e-12 * abs(float(la))
    assert (torch.tril(A) - torch.tril(B)).abs().max() <= 1e-11


@pytest.mark.parametrize("m,p_cols", [(1, []), (2, [2]), (3, [3, 4, 5])])
@pytest.mark.parametrize("n1,n2", [(5, 9), (64, 64), (130, 77), (300, 513)])
def test_kernel_input_gradients_match_oracle(env, m, p_cols, n1, n2):
    """gpar_gram_input_grad (+ the host chain from features to design-matrix columns) against the oracle's explicit
    derivative matrices: rectangular weights, and the symmetric case where both arguments of the kernel move; and the
    weighted parameter sums used by the posterior-mean gradient."""
    torch, hip, dev, to_dev = env
    from gpar_amd.engine import HipEngine
    from gpar_amd.kernels import compile_kernel
    from oracle import kernels as ok

    eng = HipEngine()
    width = m + (max(p_cols) - m + 1 if p_cols else 0)
    rng = np.random.default_rng(n1 + 3 * n2 + m)
    x1, x2 = rng.uniform(-1, 1, (n1, width)), rng.uniform(-1, 1, (n2, width))
    W = rng.standard_normal((n1, n2))
    Ws = rng.standard_normal((n1, n1))
    Ws = Ws + Ws.T
    for name, k in _kernels(m, p_cols).items():
        if name == "zero":
            continue
        ck = compile_kernel(k, width)
        spec = ok.spec_to_dict(k.resolve(width))
        got = eng.kernel_input_grads(ck, to_dev(x1), to_dev(x2), to_dev(W)).cpu().numpy()
        ref = ok.kernel_input_grads(spec, x1, x2, W)
        scale = max(1.0, np.max(np.abs(ref)))
        assert np.max(np.abs(got - ref)) <= 1e-11 * scale, name
        lower = to_dev(np.tril(Ws) + np.triu(np.full((n1, n1), np.nan), 1))
        got = eng.kernel_input_grads(ck, to_dev(x1), None, lower, sym=True).cpu().numpy()
        ref = 2.0 * ok.kernel_input_grads(spec, x1, x1, Ws)
        assert np.max(np.abs(got - ref)) <= 1e-11 * max(1.0, np.max(np.abs(ref))), name


# ---- per-specification (run-time compiled) Gram kernels --------------------------------------------------------------------

def _jit_cases():
    """(name, kernel, width): the kernel families GPARRegressor builds (gpar/regression.py:92-180) at several widths."""
    from gpar_amd.regression import GPARRegressor, _construct_gpar
    from gpar_amd.engine import set_engine
    from oracle.engine import OracleEngine

    previous = set_engine(OracleEngine())   # (only to instantiate the hyper-parameters of the host-side model objects)
    try:
        out = []
        for name, kw, m, layer in [
            ("eq-lin-eq", dict(scale=0.5, linear=True, nonlinear=True, markov=2), 4, 7),
            ("eq-only", dict(scale=0.5, linear=False), 2, 0),
            ("per-rq-wide", dict(scale=0.5, per=True, rq=True, linear=True, nonlinear=True), 3, 15),
            ("rq-inputlinear-const", dict(scale=0.7, rq=True, input_linear=True, linear=True, nonlinear=True), 2, 3),
            ("markov0-constant-term", dict(linear=True, nonlinear=True, markov=0), 1, 2),
        ]:
            reg = GPARRegressor(**kw)
            f, _ = _construct_gpar(reg, reg.vs, m, layer + 1).layers[layer]()
            out.append((name, f.kernel, m + layer))
        return out
    finally:
        set_engine(previous)


@pytest.mark.parametrize("case", range(5))
def test_generated_gram_kernel_is_bit_identical_to_the_interpreter(case, monkeypatch):
    """The kernel compiled at run time for a layer's STRUCTURE (csrc/gram_jit.h, hiprtc) and the ahead-of-time interpreter
    (csrc/gram.h) share their arithmetic verbatim (csrc/gram_math.inc): every entry must agree to the last bit - symmetric
    lower-triangular builds with noise diagonal and jitter, ragged sizes, cross-Gram with row scaling, a batch."""
    import ctypes

    import torch

    from gpar_amd import _lib
    from gpar_amd import hip as H
    from gpar_amd.kernels import compile_kernel

    name, kernel, width = _jit_cases()[case]
    dev = torch.device("cuda:0")
    ck = compile_kernel(kernel, width)
    g = torch.Generator().manual_seed(case)
    x = torch.randn(333, width, generator=g, dtype=torch.float64).to(dev)
    x2 = torch.randn(130, width, generator=g, dtype=torch.float64).to(dev)
    z, z2 = H.featurize(ck, x), H.featurize(ck, x2)
    noise = torch.rand(333, generator=g, dtype=torch.float64).to(dev)
    rs = torch.rand(333, generator=g, dtype=torch.float64).to(dev) + 0.5
    zb = H.featurize(ck, torch.randn(3 * 70, width, generator=g, dtype=torch.float64).to(dev))

    def build():
        sym = H.gram(ck, z, None, lower=True, diag_add=noise, diag_const=1e-12, out=torch.zeros(333, 336, dtype=torch.float64, device=dev)[:, :333])
        cross = H.gram(ck, z, z2, row_scale=rs)
        batch = H.gram_batch_(ck, zb, 3, H.alloc_matrix(210, 70, dev, zero=True), lower=True, diag_const=0.1)
        torch.cuda.synchronize()
        return torch.tril(sym).clone(), cross.clone(), batch.clone()

    lib = _lib.load()
    counts = lambda: tuple(c.value for c in cs) if not lib.gpar_jit_stats(*[ctypes.byref(c) for c in cs]) else None
    cs = [ctypes.c_int(), ctypes.c_int(), ctypes.c_int()]
    monkeypatch.setenv("GPAR_GRAM_JIT_MIN_ENTRIES", "-1")   # never: the interpreter
    ref = build()
    before = counts()
    monkeypatch.setenv("GPAR_GRAM_JIT_MIN_ENTRIES", "0")    # always: the generated kernel
    got = build()
    after = counts()
    assert after[1] == before[1], "a generated kernel failed to compile"
    assert after[2] >= 1   # at least this structure is cached now (42 dims: the wide form of round 5, 4 x 4 micro-tile, rolled dim loops)
    for a, b in zip(got, ref):
        assert torch.equal(a, b), name
    # and both agree with the numpy oracle's kernel evaluation
    from oracle import kernels as ok

    want = ok.gram(ok.spec_to_dict(kernel.resolve(width)), x.cpu().numpy(), x2.cpu().numpy()) * rs.cpu().numpy()[:, None]
    np.testing.assert_allclose(got[1].cpu().numpy(), want, rtol=1e-13, atol=1e-15)


def test_gram_exponential_over_its_whole_range():
    """`gram_exph8` (csrc/gram_math.inc: 64-entry table, degree 5, one-step reduction on the doubled exponent) through an EQ
    kernel on one input dim: distances chosen so that the exponent E / 2 = d^2 / (2 l^2) sweeps 0 .. 800 - values from 1 down
    through the denormals to 0.  Stated accuracy: absolute error below 2 ulp of the coefficient everywhere (entries near 1 to the
    last bit or two), relative error |k| * 8.7e-19 + rounding (k = 64 E / (2 ln 2) <= 7.4e4 before underflow), the diagonal
    exactly the coefficient, no NaN / Inf for absurd distances."""
    import mpmath as mp
    import torch

    from gpar_amd import hip as H
    from gpar_amd.kernels import compile_kernel

    name, kernel, width = _jit_cases()[1]   # scale * EQ over two input dims
    assert name == "eq-only"
    dev = torch.device("cuda:0")
    ck = compile_kernel(kernel, width)
    rng = np.random.default_rng(3)
    half = np.concatenate([[0.0], rng.uniform(0, 1, 80), rng.uniform(0, 40, 400), rng.uniform(600, 800, 150), [745.1, 746.0, 1e6, 1e14]])
    x = np.zeros((half.size, width))
    z0 = H.featurize(ck, torch.tensor([[1.0, 0.0], [0.0, 0.0]], dtype=torch.float64, device=dev)).cpu().numpy()
    per_unit = z0[0, 0] - z0[1, 0]                      # feature units per input unit (1 / length scale)
    x[:, 0] = np.sqrt(2.0 * half) / per_unit
    z = H.featurize(ck, torch.tensor(x, dtype=torch.float64, device=dev))
    K = H.gram(ck, z, None).cpu().numpy()
    coef = K[0, 0]
    assert np.isfinite(K).all() and (np.diag(K) == coef).all()
    zf = z.cpu().numpy()
    mp.mp.prec = 120
    worst_abs = worst_rel = 0.0
    for i in range(half.size):
        d0 = np.float64(zf[i, 0] - zf[0, 0])
        assert zf[i, 1] == zf[0, 1]
        e2 = mp.mpf(float(d0 * d0))   # the kernel's own rounded squared distance: what the exponential is handed
        want = mp.mpf(float(coef)) * mp.exp(-e2 / 2)
        got = mp.mpf(float(K[i, 0]))
        worst_abs = max(worst_abs, float(abs(got - want) / coef))
        if want > mp.mpf(2) ** -1000:
            worst_rel = max(worst_rel, float(abs(got - want) / want))
    assert worst_abs < 4.5e-16, worst_abs
    assert worst_rel < 1e-13, worst_rel
    assert K[-1, 0] == 0.0 and K[-2, 0] == 0.0 and K[-3, 0] >= 0.0


@pytest.mark.parametrize("case", range(5))
def test_generated_gradient_kernels_match_the_interpreter(case, monkeypatch):
    """The run-time compiled gradient passes (csrc/grad_jit.h) against the ahead-of-time interpreter (csrc/gram.h): the moment sums
    of the parameter-gradient pass with symmetric and rectangular weights (ragged sizes; with frequency derivatives where the
    kernel has periodic features) and the input-gradient pass in both modes.  Same sums in a different order: agreement to
    rounding (1e-11 of the largest sum), not to the bit."""
    import ctypes

    import torch

    from gpar_amd import _lib
    from gpar_amd import hip as H
    from gpar_amd.kernels import compile_kernel

    name, kernel, width = _jit_cases()[case]
    dev = torch.device("cuda:0")
    ck = compile_kernel(kernel, width)
    g = torch.Generator().manual_seed(100 + case)
    x = torch.randn(333, width, generator=g, dtype=torch.float64).to(dev)
    x2 = torch.randn(130, width, generator=g, dtype=torch.float64).to(dev)
    z, z2 = H.featurize(ck, x), H.featurize(ck, x2)
    periodic = any(f.periods is not None for t in ck.kernel.terms for f in t.factors)
    zd = H.featurize_dfreq(ck, x) if periodic else None
    zd2 = H.featurize_dfreq(ck, x2) if periodic else None
    Wsym = torch.randn(333, 336, generator=g, dtype=torch.float64).to(dev)[:, :333]
    Wrect = torch.randn(333, 130, generator=g, dtype=torch.float64).to(dev)

    def run():
        out = [H.gram_grad(ck, z, zd, Wsym, nblocks=7), H.gram_grad_cross(ck, z, zd, z2, zd2, Wrect.contiguous(), H.GRAD_RECT, nblocks=5),
               H.gram_grad_cross(ck, z, zd, z, zd, Wsym, H.GRAD_SYM)]
        if ck.dz:
            out += [H.gram_input_grad(ck, z, z2, Wrect.contiguous(), H.GRAD_RECT), H.gram_input_grad(ck, z, z, Wsym, H.GRAD_SYM)]
        torch.cuda.synchronize()
        return [o.clone() for o in out]

    lib = _lib.load()
    cs = [ctypes.c_int(), ctypes.c_int(), ctypes.c_int()]
    monkeypatch.setenv("GPAR_GRAD_JIT_MIN_ENTRIES", "-1")
    ref = run()
    lib.gpar_jit_stats(*[ctypes.byref(c) for c in cs])
    failures = cs[1].value
    monkeypatch.setenv("GPAR_GRAD_JIT_MIN_ENTRIES", "0")
    got = run()
    lib.gpar_jit_stats(*[ctypes.byref(c) for c in cs])
    assert cs[1].value == failures, "a generated gradient kernel failed to compile"
    for a, b in zip(got, ref):
        scale = float(b.abs().max()) + 1e-300
        assert float((a - b).abs().max()) <= 1e-11 * scale, (name, float((a - b).abs().max()), scale)


def _rank_deficient(n, ell, jitter, seed=0):
    x = np.sort(np.random.default_rng(seed).uniform(0, 1, n))
    return np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2 / ell ** 2) + jitter * np.eye(n)


@pytest.mark.parametrize("n,ell,jitter", [(2048, 0.1, 1e-12), (1024, 0.5, 1e-12), (512, 0.1, 1e-10), (314, 0.5, 1e-12), (1300, 0.3, 1e-11)])
def test_factorisation_of_a_numerically_rank_deficient_matrix_is_backward_stable(n, ell, jitter):
    """K_zz of many inducing inputs on ONE axis plus lab's jitter - what `x_ind = np.linspace(...)` gives the reference
    (gpar/model.py:286-287; examples/paper/air_temp.py) - is rank-deficient to working precision: after ~30 pivots the Schur
    complement is the jitter.  LAPACK factors it with a backward error of 2e-15.  The fused panel kernel solves its strips
    through explicit inverses of 16 x 16 diagonal blocks, which is not backward stable for the block in which the pivots fall by
    six orders of magnitude: 1e-13, and one of these cases failed with "not positive definite" - until such blocks got a step of
    iterative refinement (csrc/panel2.h: refinement flags).  Now: succeeds, and |L L^T - A| within 4x LAPACK's."""
    import torch

    from gpar_amd import hip as H

    dev = torch.device("cuda:0")
    A = _rank_deficient(n, ell, jitter)
    ref = np.linalg.cholesky(A)
    ref_err = np.abs(ref @ ref.T - A).max()
    B = H.alloc_matrix(n, n, dev)
    B.copy_(torch.tensor(A, device=dev))
    logdet, info = H.potrf_(B)
    assert int(info.item()) == 0
    L = torch.tril(B).cpu().numpy()
    err = np.abs(L @ L.T - A).max()
    assert err <= 4 * ref_err + 1e-15, (err, ref_err)
    # the log-determinant of such a matrix is a sum of ~n logs of pivots that are the jitter plus rounding noise (1e-16 n against
    # 1e-12: each determined to 3-4 digits, by LAPACK as much as here): agreement to that
    want = 2 * np.log(np.diag(ref)).sum()
    assert abs(float(logdet) - want) <= 1e-3 * abs(want), (float(logdet), want)


@pytest.mark.parametrize("n", [1024, 1300])
def test_many_row_solves_against_an_ill_conditioned_factor_are_backward_stable(n):
    """The same blocks in the fused triangular solves (`gpar_trsm_rlt`: K_xz L_z^-T of the inducing-point path, 4096 rows;
    `gpar_trsm_rln`: the backward solve of its gradient): residuals |X L^T - B| and |X L - B| relative to |X| |L| at rounding
    level, as substitution would leave them."""
    import torch

    from gpar_amd import hip as H

    dev = torch.device("cuda:0")
    L = np.linalg.cholesky(_rank_deficient(n, 0.2, 1e-12, seed=3))
    Ld = H.alloc_matrix(n, n, dev)
    Ld.copy_(torch.tensor(L, device=dev))
    rhs = np.random.default_rng(4).standard_normal((4096, n))
    for solve, apply in ((H.trsm_rlt_, lambda X: X @ L.T), (H.trsm_rln_, lambda X: X @ L)):
        X = H.alloc_matrix(4096, n, dev)
        X.copy_(torch.tensor(rhs, device=dev))
        solve(Ld, X)
        Xh = X.cpu().numpy()
        resid = np.abs(apply(Xh) - rhs).max(axis=1)
        scale = (np.abs(Xh) @ np.abs(L).sum(axis=0 if solve is H.trsm_rlt_ else 1)) + np.abs(rhs).max(axis=1)
        assert np.isfinite(Xh).all()
        assert (resid <= 1e-13 * scale).all(), float((resid / scale).max())


def test_build_time_archive_serves_mid_size_training_without_a_compilation(env):
    """A fit at n = 1100 (2^20 weight entries: above the archive's floor of 2^19, far below the run-time compilation thresholds)
    loads its Gram and gradient kernels from gpar_aot_gfx950.bin: the archive reports entries and loads, hiprtc compiles nothing."""
    import ctypes

    torch, hip, dev, to_dev = env
    from gpar_amd import _lib
    from gpar_amd.engine import HipEngine, set_engine
    from gpar_amd.regression import GPARRegressor

    lib = _lib.load()
    before = [ctypes.c_int(), ctypes.c_int(), ctypes.c_int()]
    lib.gpar_jit_stats(*[ctypes.byref(c) for c in before])
    loaded0 = ctypes.c_int()
    lib.gpar_aot_stats(None, ctypes.byref(loaded0))
    previous = set_engine(HipEngine(device="cuda:0", seed=2))
    try:
        rng = np.random.default_rng(5)
        x = rng.uniform(0, 1, (1100, 2))
        y = np.stack([np.sin(5 * x[:, 0]), np.cos(4 * x[:, 1]) + x[:, 0]], axis=1) + 0.1 * rng.standard_normal((1100, 2))
        reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1)
        reg.fit(x, y, iters=2)
    finally:
        set_engine(previous)
    after = [ctypes.c_int(), ctypes.c_int(), ctypes.c_int()]
    lib.gpar_jit_stats(*[ctypes.byref(c) for c in after])
    entries, loaded = ctypes.c_int(), ctypes.c_int()
    lib.gpar_aot_stats(ctypes.byref(entries), ctypes.byref(loaded))
    if os.environ.get("GPAR_AOT", "1") == "0" or entries.value == 0:
        pytest.skip("no kernel archive next to the library")
    assert entries.value >= 300
    assert after[0].value == before[0].value, "hiprtc compiled a kernel the archive should hold"
    assert loaded.value > loaded0.value or after[2].value > 0   # (loaded now, or already cached by an earlier test of this process)
