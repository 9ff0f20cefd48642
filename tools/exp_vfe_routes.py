"""Error of the inducing-point (VFE) bound by two routes, against conditioning of K_zz.

    route B (shipped):  Bs = D^-1/2 K_xz L_z^-T (an n x M triangular solve),  A = I + Bs^T Bs,  c = Bs^T ys
    route G (no n x M solve):  G0 = Ks^T Ks with Ks = D^-1/2 K_xz,  A = I + L_z^-1 G0 L_z^-T,  c = L_z^-1 (Ks^T ys)

Both in fp64 (numpy / LAPACK), compared with the same bound in 80-bit long double (hand-written Cholesky / substitutions).
Prints one row per case: cond(K_zz + eps I), the pivot-spread estimate (max L_jj / min L_jj)^2 the device has for free, the
relative errors of the two routes.  CPU only:  python tools/exp_vfe_routes.py
"""
import json
import sys

import numpy as np
import scipy.linalg as sl

LD = np.longdouble


def chol_ld(A):
    A = np.array(A, dtype=LD)
    n = A.shape[0]
    L = np.zeros_like(A)
    for j in range(n):
        s = A[j, j] - np.dot(L[j, :j], L[j, :j])
        L[j, j] = np.sqrt(s)
        if j + 1 < n:
            L[j + 1:, j] = (A[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    return L


def solve_lower_ld(L, B):
    B = np.array(B, dtype=LD)
    X = np.zeros_like(B)
    for i in range(L.shape[0]):
        X[i] = (B[i] - L[i, :i] @ X[:i]) / L[i, i]
    return X


def eq(a, b, scale):
    d2 = ((a[:, None, :] - b[None, :, :]) / scale) ** 2
    return np.exp(-0.5 * d2.sum(-1))


def bound(Kzz, Kxz, kdiag, d, y, route, dtype):
    n, M = Kxz.shape
    if dtype is LD:
        chol, solve = chol_ld, solve_lower_ld
    else:
        chol = lambda A: np.linalg.cholesky(A)
        solve = lambda L, B: sl.solve_triangular(L, B, lower=True)
    Kzz, Kxz, kdiag, d, y = (np.asarray(v, dtype=dtype) for v in (Kzz, Kxz, kdiag, d, y))
    Lz = chol(Kzz)
    rs = 1.0 / np.sqrt(d)
    Ks = Kxz * rs[:, None]
    ys = y * rs
    if route == "B":
        Bs = solve(Lz, Ks.T).T
        AmI = Bs.T @ Bs
        c = Bs.T @ ys
    else:
        G0 = Ks.T @ Ks
        T = solve(Lz, G0)            # L_z^-1 G0
        AmI = solve(Lz, T.T).T       # (L_z^-1 (L_z^-1 G0)^T)^T = L_z^-1 G0 L_z^-T
        AmI = 0.5 * (AmI + AmI.T)
        c = solve(Lz, Ks.T @ ys)
    A = AmI + np.eye(M, dtype=dtype)
    La = chol(A)
    q = solve(La, c)
    logdet = 2.0 * np.sum(np.log(np.diag(La)))
    trace = np.sum(kdiag / d) - np.trace(AmI)
    val = -0.5 * (trace + np.sum(np.log(d)) + n * np.log(2 * np.pi) + logdet + ys @ ys - q @ q)
    spread = float((np.max(np.diag(Lz)) / np.min(np.diag(Lz))) ** 2)
    return val, spread


def case(seed, n, M, D, scale, cluster, eps=1e-12, noise=0.1):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 1, (n, D))
    z = rng.uniform(0, 1, (M, D))
    if cluster > 0:   # pull pairs of inducing points together: the smaller the distance, the worse K_zz
        z[M // 2:] = z[: M - M // 2] + cluster * rng.standard_normal((M - M // 2, D))
    y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n)
    Kzz = eq(z, z, scale) + eps * np.eye(M)
    Kxz = eq(x, z, scale)
    kdiag = np.ones(n)
    d = np.full(n, noise)
    ref, _ = bound(Kzz, Kxz, kdiag, d, y, "B", LD)
    ref_g, _ = bound(Kzz, Kxz, kdiag, d, y, "G", LD)
    out = {"n": n, "M": M, "D": D, "scale": scale, "cluster": cluster, "cond": float(np.linalg.cond(Kzz)), "ref": float(ref)}
    for route in ("B", "G"):
        try:
            val, spread = bound(Kzz, Kxz, kdiag, d, y, route, np.float64)
            out[f"err_{route}"] = float(abs(LD(val) - ref) / abs(ref))
            out["spread"] = spread
        except np.linalg.LinAlgError:
            out[f"err_{route}"] = None
    out["ld_routes_agree"] = float(abs(ref - ref_g) / abs(ref))
    return out


def main():
    rows = []
    for D, scale in ((1, 0.3), (2, 0.5), (8, 0.5)):
        for cluster in (0.0, 3e-1, 1e-1, 3e-2, 1e-2, 3e-3, 1e-3, 1e-4, 1e-5):
            rows.append(case(1, 1500, 96, D, scale, cluster))
            r = rows[-1]
            fmt = lambda v: "   fail  " if v is None else f"{v:9.2e}"
            print(f"D={D} scale={scale} cluster={cluster:7.0e}  cond={r['cond']:9.2e} spread={r.get('spread', float('nan')):9.2e}  "
                  f"err_B={fmt(r['err_B'])}  err_G={fmt(r['err_G'])}  (long-double routes agree to {r['ld_routes_agree']:.1e})", flush=True)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
