"""Where does gpar_chol_inverse spend its time? (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpar_amd import hip, _lib
dev = torch.device("cuda:0"); lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = torch.Generator(device="cpu"); g.manual_seed(n)
X = torch.rand(n, 4, generator=g, dtype=torch.float64).to(dev)
K = hip.alloc_matrix(n, n, dev); K.copy_(torch.exp(-0.5 * torch.cdist(X, X) ** 2 / 0.25)); K.diagonal().add_(0.1)
hip.potrf_(K)
def timeit(fn, reps=2):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
print(f"chol_inverse n={n}: {timeit(lambda: hip.chol_inverse(K)):.2f} ms")
I = hip.alloc_matrix(n, n, dev)
def trsm_full():
    I.zero_(); I.diagonal().fill_(1.0); hip.trsm_rlt_(K, I)
print(f"  set identity + full trsm_rlt (no triangular skipping, n^3 flops): {timeit(trsm_full):.2f} ms")
Xi = I
out = hip.alloc_matrix(n, n, dev)
print(f"  SYRK X X^T full K (no K_FROM_ROW): {timeit(lambda: hip.gemm(Xi, Xi, tb=True, out=out, c_lower=True)):.2f} ms")
