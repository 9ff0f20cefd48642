"""Experiment: four n = 4096 augmented factorisations - one lock-step batch of four against two batches of two on two streams
(the spin chain alternates their panel launches) against four lone factorisations on four streams.  Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gpar_amd import hip as H
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4   # matrices in all
N = n + 1
g = torch.Generator().manual_seed(1)
X = torch.rand(n, 4, generator=g, dtype=torch.float64).to(dev)
K0 = H.alloc_matrix(N, N, dev, zero=True)
K0[:n, :n] = torch.exp(-0.5 * torch.cdist(X, X) ** 2 / 0.25); K0[:n, :n].diagonal().add_(0.1); K0[n, :n] = torch.sin(5 * X[:, 0])
def stacked(b):
    A = H.alloc_matrix(b * N, N, dev)
    for i in range(b): A[i * N:(i + 1) * N].copy_(K0)
    return A
def timed(fn, prep):
    best = 1e9
    for it in range(6):
        prep(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
A4 = stacked(B)
t4 = timed(lambda: H.potrf_batch_(A4, B, nf=n), lambda: [A4[i * N:(i + 1) * N].copy_(K0) for i in range(B)])
print(f"one batch of {B}: {t4:.3f} ms")
del A4; torch.cuda.empty_cache()
pool = [torch.cuda.Stream(device=dev) for _ in range(4)]
A2 = [stacked(B // 2), stacked(B // 2)]
def two():
    cur = torch.cuda.current_stream()
    for k in range(2):
        pool[k].wait_stream(cur)
        with torch.cuda.stream(pool[k]): H.potrf_batch_(A2[k], B // 2, nf=n)
    for k in range(2): cur.wait_stream(pool[k])
t2 = timed(two, lambda: [A2[k][i * N:(i + 1) * N].copy_(K0) for k in range(2) for i in range(B // 2)])
print(f"two batches of {B // 2} on two streams: {t2:.3f} ms")
if B != 4: sys.exit(0)
A1 = [stacked(1) for _ in range(4)]
def four():
    cur = torch.cuda.current_stream()
    for k in range(4):
        pool[k].wait_stream(cur)
        with torch.cuda.stream(pool[k]): H.potrf_(A1[k], nf=n)
    for k in range(4): cur.wait_stream(pool[k])
t1 = timed(four, lambda: [A1[k].copy_(K0) for k in range(4)])
print(f"four lone on four streams: {t1:.3f} ms")
