"""Gram build of the widest C3 layer kernel at n = 16384 (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpar_amd import hip
from gpar_amd.kernels import EQ, Linear, compile_kernel
dev = torch.device("cuda:0")
n = 16384
x = torch.rand(n, 6, dtype=torch.float64, device=dev)
k = (1.0 * EQ().stretch(np.full(4, 0.5))).select([0, 1, 2, 3]) + (Linear().stretch(np.full(2, 100.0)) + 1.0 * EQ().stretch(np.ones(2))).select([4, 5])
ck = compile_kernel(k, 6)
z = hip.featurize(ck, x)
A = hip.alloc_matrix(n, n, dev)
for _ in range(2): hip.gram(ck, z, None, out=A, lower=True, diag_const=0.1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): hip.gram(ck, z, None, out=A, lower=True, diag_const=0.1)
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"gram n={n} (EQ4 + Linear2 + EQ2, lower): {ms:.3f} ms  {8*n*(n+1)/2/ms*1e-6:.0f} GB/s written")
