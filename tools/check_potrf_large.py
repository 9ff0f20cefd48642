"""Size-independent correctness property of gpar_potrf at sizes the oracle cannot reach: L (L^T v) = K v."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpar_amd import hip
dev = torch.device("cuda:0")
for n in [int(a) for a in sys.argv[1:]] or [32768]:
    g = torch.Generator(device="cpu"); g.manual_seed(n)
    X = torch.rand(n, 4, generator=g, dtype=torch.float64).to(dev)
    K = hip.alloc_matrix(n, n, dev)
    K.copy_(torch.exp(-0.5 * torch.cdist(X, X) ** 2 / 0.25)); K.diagonal().add_(0.1)
    v = torch.randn(n, 3, dtype=torch.float64, generator=g).to(dev)
    Kv = K[:, :n] @ v
    A = hip.alloc_matrix(n, n, dev); A.copy_(K)   # (K.clone() would be contiguous: an odd leading dimension takes the scalar paths)
    logdet = torch.zeros(1, dtype=torch.float64, device=dev); info = torch.zeros(1, dtype=torch.int32, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); hip.potrf_(A, logdet=logdet, info=info); e1.record(); e1.synchronize()
    L = torch.tril(A[:, :n])
    err = ((L @ (L.T @ v)) - Kv).abs().max().item() / Kv.abs().max().item()
    print(f"n={n}: potrf {e0.elapsed_time(e1):.1f} ms ({n**3/3/e0.elapsed_time(e1)*1e-9:.1f} TFLOP/s), info={int(info.item())}, "
          f"|L L^T v - K v| / |K v| = {err:.2e}, logdet={logdet.item():.6f}")
    del K, A, L, X
