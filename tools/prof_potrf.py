import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpar_amd import hip
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = torch.Generator(device="cpu"); g.manual_seed(n)
X = torch.rand(n, 4, generator=g, dtype=torch.float64).to(dev)
K0 = hip.alloc_matrix(n, n, dev)
K0.copy_(torch.exp(-0.5 * torch.cdist(X, X) ** 2 / 0.25)); K0.diagonal().add_(0.1)
A = hip.alloc_matrix(n, n, dev)
for it in range(3):
    A.copy_(K0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); hip.potrf_(A); e1.record(); e1.synchronize()
    print(f"potrf n={n}: {e0.elapsed_time(e1):.2f} ms")
