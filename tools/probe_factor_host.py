import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd import hip
eng = HipEngine(seed=1); set_engine(eng)
n = 4096
dev = eng.device
K = hip.alloc_matrix(n, n, dev); 
x = torch.rand(n, 3, dtype=torch.float64, device=dev)
K.copy_(torch.exp(-0.5 * torch.cdist(x, x) ** 2 / 0.25)); K.diagonal().add_(0.1)
rhs = torch.randn(n, 1, dtype=torch.float64, device=dev)
def T(): return time.perf_counter()
for rep in range(3):
    torch.cuda.synchronize()
    marks = []
    for layer in range(3):
        t0 = T(); A = eng.new_matrix(n + 1, n + 1); t1 = T()
        A[:n, :n].copy_(K); t2 = T()
        A[n, :n] = rhs.reshape(-1); t3 = T()
        A[n, n] = 0.0; t4 = T()
        logdet, info = eng.potrf_(A, nf=n); t5 = T()
        q = -A[n, n]; t6 = T()
        marks.append([round(1e3 * (b - a), 3) for a, b in zip([t0, t1, t2, t3, t4, t5], [t1, t2, t3, t4, t5, t6])])
    torch.cuda.synchronize()
    print("alloc copyK rowset cornerset potrf neg (ms):", marks)
