"""TN against NT at the inducing-point shapes (development aid): A = Bs^T Bs with Bs 65536 x 1024 (split-K, lower) as the product
stores it (TN: k-major operand) and from a transposed copy (NT: k contiguous); plus plain 8192^3 / 2 TN vs NT."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpar_amd import hip
dev = torch.device("cuda:0")
def best(run, reps=6):
    run(); torch.cuda.synchronize(); b = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); e1.synchronize(); b = min(b, e0.elapsed_time(e1))
    return b
n, M = 65536, 1024
Bs = hip.alloc_matrix(n, M, dev); Bs.copy_(torch.randn(n, M, dtype=torch.float64, device=dev))
Bt = hip.alloc_matrix(M, n, dev); Bt.copy_(Bs.t())
C1 = hip.alloc_matrix(M, M, dev); C2 = hip.alloc_matrix(M, M, dev)
flops = 36 * 128 * 128 * n * 2.0   # lower tiles actually computed
t = best(lambda: hip.gemm(Bs, Bs, ta=True, out=C1, c_lower=True))
print(f"TN  Bs^T Bs (k-major operand, as stored): {t:.3f} ms  {flops / t * 1e-9:.1f} TFLOP/s")
t = best(lambda: hip.gemm(Bt, Bt, tb=True, out=C2, c_lower=True))
print(f"NT  Bt Bt^T (k contiguous):               {t:.3f} ms  {flops / t * 1e-9:.1f} TFLOP/s   max diff {float((torch.tril(C1) - torch.tril(C2)).abs().max()):.2e}")
for n2 in (8192,):
    A = torch.randn(n2, n2, dtype=torch.float64, device=dev); out = hip.alloc_matrix(n2, n2, dev)
    for name, kw in (("NN", {}), ("NT", dict(tb=True)), ("TN", dict(ta=True)), ("TT", dict(ta=True, tb=True))):
        t = best(lambda: hip.gemm(A, A, out=out, **kw), 3)
        print(f"{name} {n2}^3: {t:.3f} ms  {2.0 * n2 ** 3 / t * 1e-9:.1f} TFLOP/s")
