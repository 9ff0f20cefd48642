"""gpar_potrf wall time at the given sizes (augmented (n + 1) matrix, best of 5); development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpar_amd import hip
dev = torch.device("cuda:0")
for n in [int(a) for a in sys.argv[1:]] or [16384]:
    g = torch.Generator(device="cpu"); g.manual_seed(n)
    X = torch.rand(n, 4, generator=g, dtype=torch.float64).to(dev)
    K0 = hip.alloc_matrix(n + 1, n + 1, dev, zero=True)
    K0[:n, :n] = torch.exp(-0.5 * torch.cdist(X, X) ** 2 / 0.25); K0[:n, :n].diagonal().add_(0.1)
    K0[n, :n] = torch.sin(5 * X[:, 0])
    A = hip.alloc_matrix(n + 1, n + 1, dev)
    best = 1e9
    for it in range(5):
        A.copy_(K0); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); logdet, info = hip.potrf_(A, nf=n); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    assert int(info.item()) == 0
    print(f"potrf n={n}: {best:.3f} ms  {n**3/3/best*1e-9:.2f} TFLOP/s  logdet {float(logdet):.12e} checksum {float(torch.tril(A).sum()):.12e}", flush=True)
    del K0, A, X; torch.cuda.empty_cache()
