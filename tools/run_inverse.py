"""gpar_chol_inverse alone, for profiling:  python tools/run_inverse.py [n] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpar_amd import hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu"); g.manual_seed(n)
X = torch.rand(n, 4, generator=g, dtype=torch.float64).to(dev)
K = hip.alloc_matrix(n, n, dev); K.copy_(torch.exp(-0.5 * torch.cdist(X, X) ** 2 / 0.25)); K.diagonal().add_(0.1)
hip.potrf_(K)
torch.cuda.synchronize()
for _ in range(reps):
    time.sleep(0.05)
    t0 = time.perf_counter(); out = hip.chol_inverse(K); torch.cuda.synchronize()
    print(f"chol_inverse n={n}: {1e3 * (time.perf_counter() - t0):.2f} ms")
