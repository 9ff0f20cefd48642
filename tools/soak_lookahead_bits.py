import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gpar_amd import hip
dev = torch.device("cuda:0")
for n, ell, jit in ((5200, 0.2, 1e-10), (5200, 0.2, 0.1), (5200, 0.2, 1e-6), (8192, 0.2, 1e-10)):
    x = np.sort(np.random.default_rng(0).uniform(0, 1, n))
    A0 = torch.tensor(np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2 / ell ** 2) + jit * np.eye(n), device=dev)
    res = {}
    for la in (True, False):
        outs = []
        for r in range(6):
            B = hip.alloc_matrix(n, n, dev); B.copy_(A0)
            logdet, info = hip.potrf_(B, lookahead=la)
            outs.append(torch.tril(B).clone())
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        res[la] = outs[0]
        print(n, jit, "lookahead", la, "info", int(info.item()), "repeatable:", same, flush=True)
    d = (res[True] - res[False]).abs()
    idx = torch.nonzero(d > 0)
    print("   on vs off: max diff %.3e, first differing entry %s, count %d" % (float(d.max()), idx[0].tolist() if len(idx) else None, len(idx)), flush=True)
