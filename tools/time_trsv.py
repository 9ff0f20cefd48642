"""alpha^T = z^T L^-1 (gpar_trsm_rln with one row) at n = 16384 (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpar_amd import hip
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = torch.Generator(device="cpu"); g.manual_seed(n)
X = torch.rand(n, 4, generator=g, dtype=torch.float64).to(dev)
K = hip.alloc_matrix(n, n, dev); K.copy_(torch.exp(-0.5 * torch.cdist(X, X) ** 2 / 0.25)); K.diagonal().add_(0.1)
hip.potrf_(K)
for rows in (1, 4, 64, 1024):
    b0 = torch.randn(rows, n, dtype=torch.float64, generator=g).to(dev)
    b = hip.alloc_matrix(rows, n, dev)
    def run():
        b.copy_(b0); hip.trsm_rln_(K, b)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); e1.synchronize()
    Lt = torch.tril(K[:n, :n])
    res = (b[:, :n] @ Lt - b0).abs().max().item() / b0.abs().max().item()
    print(f"trsm_rln rows={rows} n={n}: {e0.elapsed_time(e1):.2f} ms   residual {res:.2e}")
