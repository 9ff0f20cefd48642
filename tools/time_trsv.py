"""Single-row backward solve x L = b (gpar_trsm_rln with one row: the register-resident TRSV) at several n: time, bytes of L read
(8 n (n + 1) / 2) per second, residual (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpar_amd import hip
dev = torch.device("cuda:0")
for n in [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 8192, 16384, 20011]:
    g = torch.Generator().manual_seed(n)
    L = hip.alloc_matrix(n, n, dev)
    L.copy_(torch.tril(torch.rand(n, n, generator=g, dtype=torch.float64) * 0.01).to(dev))
    L.diagonal().add_(1.0)
    b0 = torch.randn(1, n, generator=g, dtype=torch.float64).to(dev)
    best = 1e9
    for _ in range(8):
        b = hip.alloc_matrix(1, n, dev); b.copy_(b0); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); hip.trsm_rln_(L, b); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    resid = float((b @ torch.tril(L) - b0).abs().max() / b.abs().max())
    print(f"trsv n={n}: {best * 1e3:8.1f} us  {8.0 * n * (n + 1) / 2 / best * 1e-9:6.3f} TB/s  residual {resid:.1e}  checksum {float(b.sum()):.12e}", flush=True)
