"""SYRK-shaped GEMM timing (development aid): C(n x n, lower) -= P P^T with K = k."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpar_amd import hip

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ks = [int(a) for a in sys.argv[2:]] or [256]
reps = int(os.environ.get("REPS", "5"))
A = hip.alloc_matrix(n, n, dev); A.zero_()
P = torch.randn(n, max(ks), dtype=torch.float64, device=dev)
for k in ks:
    Pk = P[:, :k]
    hip.gemm(Pk, Pk, tb=True, alpha=-1.0, beta=1.0, out=A, c_lower=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        hip.gemm(Pk, Pk, tb=True, alpha=-1.0, beta=1.0, out=A, c_lower=True)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"syrk n={n} k={k}: {ms:.3f} ms  {n*(n+1)*k/ms*1e-9:.2f} TFLOP/s")
