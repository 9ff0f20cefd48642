"""Dev aid: bitwise repeatability of gpar_potrf on an augmented GP matrix of the C3 benchmark (where the first differing
entry is, if any)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gpar_amd import hip
from bench import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
x, y = synthetic(n, 4, 8)
X = torch.tensor(np.concatenate([x, y[:, :2]], axis=1), device="cuda")
d = torch.cdist(X / 0.7, X / 0.7)
A0 = hip.alloc_matrix(n + 1, n + 1, "cuda", zero=True)   # padded leading dimension, as the product allocates
A0[:n, :n] = torch.exp(-0.5 * d * d) + X @ X.T * 0.01 + 0.1 * torch.eye(n, dtype=torch.float64, device="cuda")
A0[n, :n] = torch.tensor(y[:, 2], device="cuda")
del d
ref = None
nfail = 0
for la in (True,):
    for rep in range(reps):
        B = hip.alloc_matrix(n + 1, n + 1, "cuda")
        B.copy_(A0)
        out = hip.potrf_(B, nf=n, lookahead=la)
        torch.cuda.synchronize()
        if ref is None:
            ref = B.clone()
            continue
        diff = torch.tril(B) != torch.tril(ref)
        nd = int(diff.sum())
        if nd:
            nfail += 1
            idx = diff.nonzero()
            cols = idx[:, 1]
            c0 = int(cols.min())
            rows_c0 = idx[cols == c0][:, 0]
            t0 = c0 // 64 * 64
            blk = (B[t0:t0 + 64, t0:t0 + 64] - ref[t0:t0 + 64, t0:t0 + 64]).cpu().numpy()
            rat = (B[t0:t0 + 64, t0:t0 + 64] / ref[t0:t0 + 64, t0:t0 + 64]).cpu().numpy()
            np.set_printoptions(linewidth=250, precision=3)
            jb = (c0 - t0) // 8
            print('   diag tile at', t0, 'block', jb, ': rows x 8 columns of the block, got/ref - 1:')
            for r in range(8 * jb, 64):
                print('    row %2d' % r, rat[r, 8 * jb:8 * jb + 8] - 1.0)
            print(f"lookahead={la} rep={rep}: {nd} entries differ; first column {c0} (panel {c0 // 512}, tile col {c0 // 64 % 8}, col in tile {c0 % 64}), rows there {int(rows_c0.min())}..{int(rows_c0.max())} ({len(rows_c0)}), max abs {float((torch.tril(B)-torch.tril(ref)).abs().max()):.3e}")
print("done:", nfail, "of", reps, "runs differ")
