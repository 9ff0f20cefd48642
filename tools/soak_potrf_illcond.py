"""Soak: factorisations of numerically rank-deficient matrices (refined strips, csrc/panel2.h) repeated, bits compared (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpar_amd import hip
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for n, ell, jit in ((2048, 0.1, 1e-12), (1300, 0.3, 1e-11), (5200, 0.2, 1e-10)):
    x = np.sort(np.random.default_rng(0).uniform(0, 1, n))
    A0 = torch.tensor(np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2 / ell ** 2) + jit * np.eye(n), device=dev)
    ref = None
    for r in range(reps):
        B = hip.alloc_matrix(n, n, dev); B.copy_(A0)
        logdet, info = hip.potrf_(B)   # (look-ahead on / off round a ragged last tile row differently: tools/soak_lookahead_bits.py)
        assert int(info.item()) == 0, (n, r, int(info.item()))
        Lr = torch.tril(B)
        if ref is None: ref = Lr.clone()
        elif not torch.equal(Lr, ref):
            print(f"n={n}: repetition {r} differs by {float((Lr - ref).abs().max()):.3e}"); sys.exit(1)
    print(f"n={n} jitter={jit:g}: {reps} identical factorisations", flush=True)
