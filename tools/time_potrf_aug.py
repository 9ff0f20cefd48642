"""Dev aid: cost of the augmented row.  gpar_potrf on n x n against the (n + 1) x (n + 1) matrix [[K, .], [y^T, c]] with nf = n
(what every exact-GP layer evaluation factors)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpar_amd import hip

dev = torch.device("cuda:0")


def timeit(fn, reps=4):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


for n in [int(a) for a in sys.argv[1:]] or [8192, 16384]:
    g = torch.Generator(device="cpu"); g.manual_seed(n)
    X = torch.rand(n, 4, generator=g, dtype=torch.float64).to(dev)
    K0 = hip.alloc_matrix(n + 1, n + 1, dev)
    K0.zero_()
    K0[:n, :n].copy_(torch.exp(-0.5 * torch.cdist(X, X) ** 2 / 0.25))
    K0[:n, :n].diagonal().add_(0.1)
    K0[n, :n] = torch.randn(n, generator=g, dtype=torch.float64).to(dev)
    K0[n, n] = 0.0
    A = hip.alloc_matrix(n + 1, n + 1, dev)
    B = hip.alloc_matrix(n, n, dev)

    def run_aug():
        A.copy_(K0)
        hip.potrf_(A, nf=n)

    def run_plain():
        B.copy_(K0[:n, :n])
        hip.potrf_(B)

    def copy_aug():
        A.copy_(K0)

    def copy_plain():
        B.copy_(K0[:n, :n])

    ta, tp, ca, cp = timeit(run_aug), timeit(run_plain), timeit(copy_aug), timeit(copy_plain)
    print(f"n={n}: plain {tp - cp:.2f} ms, augmented (nf=n of n+1) {ta - ca:.2f} ms")
