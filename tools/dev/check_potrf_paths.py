"""Development aid: gpar_potrf / gpar_potrf_batch against numpy's Cholesky, element by element, repeated (races show as run-to-run differences)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gpar_amd import hip

dev = torch.device("cuda:0")
N, nf, batch = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (2048, 1024, 5)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
rng = np.random.default_rng(N + batch)
mats = []
for b in range(batch):
    G = rng.standard_normal((N, N + 8))
    mats.append(G @ G.T / N + (0.5 + b) * np.eye(N))
refs = [np.linalg.cholesky(m[:nf, :nf]) for m in mats]
X21 = [np.linalg.solve(refs[b], mats[b][nf:, :nf].T).T for b in range(batch)]


def report(tag, got, b):
    L = np.tril(got[:nf, :nf])
    e1 = np.abs(L - refs[b]) / np.abs(refs[b]).max()
    e2 = np.abs(got[nf:, :nf] - X21[b]) / max(1e-300, np.abs(X21[b]).max()) if N > nf else np.zeros((1, 1))
    worst = np.unravel_index(np.argmax(e1), e1.shape)
    worst2 = np.unravel_index(np.argmax(e2), e2.shape)
    print(f"{tag} matrix {b}: max err L {e1.max():.2e} at {worst}, X21 {e2.max():.2e} at ({worst2[0] + nf}, {worst2[1]})", flush=True)
    if e1.max() > 1e-12:
        bad = np.argwhere(e1 > 1e-12)
        print("   bad L entries: rows", bad[:, 0].min(), "..", bad[:, 0].max(), "cols", bad[:, 1].min(), "..", bad[:, 1].max(), "count", len(bad))
        nt = nf // 64
        tmap = e1[:nt * 64, :nt * 64].reshape(nt, 64, nt, 64).max(axis=(1, 3))
        ti, tj = [int(v) for v in np.argwhere(tmap > 1e-12)[np.argmin([a * 1000 + b for b, a in np.argwhere(tmap > 1e-12)])]]
        tj = int(np.argwhere(tmap > 1e-12)[:, 1].min())
        ti = int(np.argwhere(tmap[:, tj] > 1e-12).min())
        print(f"   first bad tile column {tj}; bad tiles in it: {np.argwhere(tmap[:, tj] > 1e-12).ravel().tolist()}")
        for tii in np.argwhere(tmap[:, tj] > 1e-12).ravel().tolist()[:2]:
            blk = e1[64 * tii:64 * tii + 64, 64 * tj:64 * tj + 64].reshape(8, 8, 8, 8).max(axis=(1, 3))
            if tii == tj:
                br, bc = [int(v) for v in np.argwhere(blk == blk.max())[0]]
                r0_, c0_ = 64 * tii + 8 * br, 64 * tj + 8 * bc
                g8, t8 = L[r0_:r0_ + 8, c0_:c0_ + 8], refs[b][r0_:r0_ + 8, c0_:c0_ + 8]
                print(f"   worst 8x8 block ({br},{bc}) of the diagonal tile: got / true - 1:")
                for r in range(8):
                    print("     ", " ".join(f"{g8[r, c] / t8[r, c] - 1:10.2e}" for c in range(8)))
            print(f"   tile ({tii},{tj}) 8x8-block max errors (log10):")
            for r in range(8):
                print("     ", " ".join(f"{np.log10(max(v, 1e-17)):6.1f}" for v in blk[r]))
    if e2.max() > 1e-12:
        bad = np.argwhere(e2 > 1e-12)
        print("   bad X21 entries: rows", bad[:, 0].min() + nf, "..", bad[:, 0].max() + nf, "cols", bad[:, 1].min(), "..", bad[:, 1].max(), "count", len(bad))


for rep in range(reps):
    stacked = hip.alloc_matrix(batch * N, N, dev)
    stacked.copy_(torch.from_numpy(np.concatenate(mats, axis=0)).to(dev))
    logdet, info = hip.potrf_batch_(stacked, batch, nf)
    got = stacked.cpu().numpy()
    for b in range(batch):
        report(f"rep {rep} batch ", got[b * N:(b + 1) * N], b)
    for b in range(min(batch, 2)):
        single = torch.from_numpy(mats[b]).to(dev).clone()
        A = hip.alloc_matrix(N, N, dev)
        A.copy_(single)
        hip.potrf_(A, nf)
        report(f"rep {rep} single", A.cpu().numpy(), b)
