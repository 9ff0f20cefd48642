"""Seed 507 of tools/fuzz_more.py: record every matrix handed to hip.potrf_ and test each afterwards: fused vs unfused vs LAPACK."""
import os, sys
sys.path.insert(0, os.getcwd())
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 507
sys.argv = ["x", "0", "0"]
exec(open("tools/fuzz_more.py").read().split("bad = 0")[0])
import numpy as np, torch
from gpar_amd import hip as H
from tests.test_fuzz_parity_gpu import _grads
saved = []
orig = H.potrf_
def spy(A, nf=None, **kw):
    saved.append((A.detach().clone(), nf))
    return orig(A, nf=nf, **kw)
H.potrf_ = spy
kw, x, y, w, xs = case(seed)
try:
    _grads("hip", kw, x, y, w)
    print("no failure")
except Exception as e:
    print("failed:", type(e).__name__, str(e)[:80])
H.potrf_ = orig
out = {}
for i, (A, nf) in enumerate(saved):
    N = A.shape[0]; nf = N if nf is None else nf
    res = []
    for fused in (True, False):
        B = H.alloc_matrix(N, N, A.device); B.copy_(A)
        logdet, info = orig(B, nf=nf, fused=fused, lookahead=False)
        res.append((int(info.item()), float(logdet)))
    K = torch.tril(A[:nf, :nf]).cpu().numpy(); K = K + np.tril(K, -1).T
    try:
        L = np.linalg.cholesky(K); lap = (0, 2 * np.log(np.diag(L)).sum(), float(np.diag(L).min()))
    except np.linalg.LinAlgError:
        lap = (1, float("nan"), float("nan"))
    if res[0][0] != 0 and not out:
        out["K"] = K
        np.save("gpurun_out/r04_notpd_matrix.npy", K)
    try:
        ev = np.linalg.eigvalsh(K)
    except np.linalg.LinAlgError:
        ev = [float("nan")] * 2
    print(i, "N", N, "nf", nf, "fused", res[0], "unfused", res[1], "lapack", lap, "eig min %.3e max %.3e" % (ev[0], ev[-1]), flush=True)
