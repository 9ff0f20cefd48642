"""Trailing-update shapes of gpar_potrf at n = 16384 through gpar_gemm (lower trapezoid, NT, beta = 1): K = 512 / 1536, several sizes (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpar_amd import hip
dev = torch.device("cuda:0")
for n in (15872, 10240, 6144):
    for K in (512, 1536):
        P = torch.randn(n, K, dtype=torch.float64, device=dev)
        C = hip.alloc_matrix(n, n, dev); C.zero_()
        run = lambda: hip.gemm(P, P, tb=True, alpha=-1.0, beta=1.0, out=C, c_lower=True)
        run(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1))
        print(f"syrk n={n} K={K}: {best:.3f} ms  {n * (n + 1) * K / best * 1e-9:.2f} TFLOP/s  checksum {float(torch.tril(C).sum()):.10e}", flush=True)
        del P, C
