"""Soak: the single-row solve repeated many times at several sizes (ragged ones included), bits compared with the first run and the
residual checked - the block kernel's waves hand off through LDS words, a race would show as a differing bit (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpar_amd import hip
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
for n in (64, 100, 129, 511, 513, 1000, 1024, 1300, 2048, 4097):
    g = torch.Generator().manual_seed(n)
    L = hip.alloc_matrix(n, n, dev)
    L.copy_(torch.tril(torch.rand(n, n, generator=g, dtype=torch.float64) * 0.01).to(dev))
    L.diagonal().add_(1.0)
    b0 = torch.randn(1, n, generator=g, dtype=torch.float64).to(dev)
    ref = None
    side = torch.cuda.Stream()
    noise = torch.randn(2048, 2048, dtype=torch.float64, device=dev)
    for r in range(reps):
        b = hip.alloc_matrix(1, n, dev); b.copy_(b0)
        if r % 3 == 1:   # something else on the chip
            with torch.cuda.stream(side):
                noise @ noise
        hip.trsm_rln_(L, b)
        if ref is None:
            ref = b.clone()
            assert float((b @ torch.tril(L) - b0).abs().max() / b.abs().max()) < 1e-12
        elif not torch.equal(b, ref):
            print(f"n={n}: repetition {r} differs by {float((b - ref).abs().max()):.3e}"); sys.exit(1)
    torch.cuda.synchronize()
    print(f"n={n}: {reps} identical solves", flush=True)
