import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpar_amd import hip
dev = torch.device("cuda:0")
n = 16384
A = hip.alloc_matrix(n, n, dev); A.zero_()
P = torch.randn(n, 512, dtype=torch.float64, device=dev)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for lower in (True, False):
    for beta in (1.0, 0.0):
        for k in (16, 32, 64, 256):
            Pk = P[:, :k]
            ms = t(lambda: hip.gemm(Pk, Pk, tb=True, alpha=-1.0, beta=beta, out=A, c_lower=lower))
            print(f"lower={lower} beta={beta} k={k}: {ms:.3f} ms")
# pure read-modify-write of the same bytes with torch for reference
ms = t(lambda: A.mul_(1.0000001))
print(f"torch A.mul_ (full n^2 r+w): {ms:.3f} ms")
