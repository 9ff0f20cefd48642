"""Quick potrf / gemm timing on the GPU box (development aid)."""
import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpar_amd import hip, _lib

dev = torch.device("cuda:0")
lib = _lib.load()

def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best

for n in [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384]:
    g = torch.Generator(device="cpu"); g.manual_seed(n)
    X = torch.rand(n, 4, generator=g, dtype=torch.float64).to(dev)
    # SPD: EQ gram + noise
    d2 = torch.cdist(X, X) ** 2
    K0 = hip.alloc_matrix(n, n, dev)
    K0.copy_(torch.exp(-0.5 * d2 / 0.25)); del d2
    K0.diagonal().add_(0.1)
    A = hip.alloc_matrix(n, n, dev)
    def run():
        A.copy_(K0)
        hip.potrf_(A)
    def copy_only():
        A.copy_(K0)
    t_all = timeit(run); t_copy = timeit(copy_only)
    t = t_all - t_copy
    print(f"potrf n={n}: {t:.2f} ms  {n**3/3/t*1e-9:.2f} TFLOP/s (copy {t_copy:.2f} ms)")
    # plain syrk-shaped gemm: C(n x n lower) -= P P^T, K = 256
    P = torch.randn(n, 256, dtype=torch.float64, device=dev)
    for kk in (64, 128, 256):
        Pk = P[:, :kk]
        tg = timeit(lambda: hip.gemm(Pk, Pk, tb=True, alpha=-1.0, beta=1.0, out=A, c_lower=True))
        print(f"  syrk n={n} k={kk}: {tg:.3f} ms  {n*(n+1)*kk/tg*1e-9:.2f} TFLOP/s")
    B = torch.randn(n, 1024, dtype=torch.float64, device=dev)
    Cc = hip.alloc_matrix(n, 1024, dev)
    tg = timeit(lambda: hip.gemm(A, B, out=Cc))
    print(f"  gemm NN {n}x1024x{n}: {tg:.3f} ms  {2*n*n*1024/tg*1e-9:.2f} TFLOP/s")
    del K0, A, P, B, Cc
