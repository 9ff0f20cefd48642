"""What the vendor library reaches on the same shapes (torch fp64 matmul -> rocBLAS / hipBLASLt), for context only:
the product never calls it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpar_amd import hip
dev = torch.device("cuda:0")
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
for (m, n, k) in [(8192, 8192, 8192), (16384, 16384, 512), (16384, 16384, 2048)]:
    A = torch.randn(m, k, dtype=torch.float64, device=dev); B = torch.randn(n, k, dtype=torch.float64, device=dev)
    C = torch.randn(m, n, dtype=torch.float64, device=dev)
    t = timeit(lambda: torch.addmm(C, A, B.T, beta=1.0, alpha=-1.0, out=C))
    print(f"rocBLAS via torch.addmm  C({m}x{n}) -= A B^T, K={k}: {t:.3f} ms  {2*m*n*k/t*1e-9:.1f} TFLOP/s (full square)")
    Cm = hip.alloc_matrix(m, n, dev); Cm.copy_(C)
    t2 = timeit(lambda: hip.gemm(A, B, tb=True, alpha=-1.0, beta=1.0, out=Cm))
    print(f"gpar_gemm (this library), same call:                       {t2:.3f} ms  {2*m*n*k/t2*1e-9:.1f} TFLOP/s")
    if m == n:
        t3 = timeit(lambda: hip.gemm(A, A, tb=True, alpha=-1.0, beta=1.0, out=Cm, c_lower=True))
        print(f"gpar_gemm lower-only (SYRK shape):                         {t3:.3f} ms  {m*(n+1)*k/t3*1e-9:.1f} TFLOP/s")
    del A, B, C, Cm
