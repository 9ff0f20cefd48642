"""Cholesky of numerically rank-deficient kernel matrices (K_zz of many inducing inputs on a 1-D axis + a small jitter): does the
factorisation succeed, and what is its backward error |L L^T - A| / |A|, for numpy (LAPACK), the fused panel path and the
unfused (substitution) path?   python tools/diag_illcond_potrf.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gpar_amd import hip as H

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for n in (314, 512, 1024, 2048):
    x = np.sort(rng.uniform(0, 1, n))
    for ell in (0.5, 0.1):
        K = np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2 / ell ** 2)
        for jit in (1e-6, 1e-8, 1e-10, 1e-12):
            A = K + jit * np.eye(n)
            row = [f"n={n} ell={ell} jitter={jit:g}"]
            try:
                L = np.linalg.cholesky(A)
                row.append("lapack ok %.1e" % (np.abs(L @ L.T - A).max()))
            except np.linalg.LinAlgError:
                row.append("lapack FAIL")
            for name, fused in (("fused", True), ("unfused", False)):
                B = H.alloc_matrix(n, n, dev)
                B.copy_(torch.tensor(A, device=dev))
                _, info = H.potrf_(B, fused=fused)
                i = int(info.item())
                if i == 0:
                    Lg = torch.tril(B).cpu().numpy()
                    row.append("%s ok %.1e" % (name, np.abs(Lg @ Lg.T - A).max()))
                else:
                    row.append("%s FAIL@%d" % (name, i))
            print("  ".join(row), flush=True)
