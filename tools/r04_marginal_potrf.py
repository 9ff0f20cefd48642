"""Factorisations at the edge of numerical definiteness (K_zz + 1e-12 of many inducing inputs on one axis): how often does each
path report a non-positive pivot where LAPACK does not, and what is its backward error?  usage: [library.so]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gpar_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from gpar_amd import hip as H

def lapack_ok(K):
    try:
        np.linalg.cholesky(K); return True
    except np.linalg.LinAlgError:
        return False

def run(K, fused):
    N = K.shape[0]
    B = H.alloc_matrix(N, N, torch.device("cuda")); B.copy_(torch.from_numpy(K).cuda())
    logdet, info = H.potrf_(B, nf=N, fused=fused, lookahead=False)
    info = int(info.item())
    if info != 0: return info, float("nan")
    L = torch.tril(B[:N, :N]).cpu().numpy().astype(np.longdouble)
    R = L @ L.T - K.astype(np.longdouble)
    return 0, float(np.abs(R).max() / np.abs(K).max())

cases = [("seed507", np.load("tools/data/r04_notpd_matrix.npy"))]
rng = np.random.default_rng(0)
for i in range(60):
    M = int(rng.integers(100, 520)); ell = float(rng.uniform(0.2, 1.5)); var = float(10 ** rng.uniform(-1, 2.7))
    z = np.sort(rng.uniform(0, rng.uniform(2, 12), M))
    K = var * np.exp(-0.5 * (z[:, None] - z[None, :]) ** 2 / ell ** 2)
    if i % 3 == 0: K = K + 0.3 * var * np.outer(z, z) / z.max() ** 2
    cases.append((f"M{M}_l{ell:.2f}_v{var:.1f}", K + 1e-12 * np.eye(M)))
tally = {"fused_only_fail": 0, "unfused_only_fail": 0, "both_fail_lapack_ok": 0, "lapack_fail": 0, "all_ok": 0}
worst = {True: 0.0, False: 0.0}
for name, K in cases:
    lok = lapack_ok(K)
    (fi, fe), (ui, ue) = run(K, True), run(K, False)
    if fi == 0: worst[True] = max(worst[True], fe)
    if ui == 0: worst[False] = max(worst[False], ue)
    key = "lapack_fail" if not lok else "all_ok" if fi == 0 and ui == 0 else "both_fail_lapack_ok" if fi and ui else "fused_only_fail" if fi else "unfused_only_fail"
    tally[key] += 1
    if key != "all_ok" or name == "seed507":
        print(f"{name:28s} lapack {'ok' if lok else 'FAIL'}  fused info {fi} err {fe:.2e}  unfused info {ui} err {ue:.2e}", flush=True)
print("tally", tally, "worst backward error fused %.2e unfused %.2e" % (worst[True], worst[False]))
