"""Do concurrent factorisations slow one another down, and does giving each its own set of compute units help?
k factorisations of n x n on k streams at once: plain streams vs streams created with disjoint CU masks
(hipExtStreamCreateWithCUMask).   python tools/probe_concurrent_potrf.py [n]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from gpar_amd import hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
x = torch.rand(n, 3, dtype=torch.float64, device=dev)
K = hip.alloc_matrix(n, n, dev)
K.copy_(torch.exp(-0.5 * torch.cdist(x, x) ** 2 / 0.25))
K.diagonal().add_(0.1)
hiprt = ctypes.CDLL("libamdhip64.so")


def masked_stream(lo, hi, total=256):
    """A stream restricted to CUs [lo, hi) (CU index as the runtime numbers them)."""
    words = (total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for cu in range(lo, hi):
        mask[cu // 32] |= 1 << (cu % 32)
    s = ctypes.c_void_p()
    rc = hiprt.hipExtStreamCreateWithCUMask(ctypes.byref(s), words, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def run(streams, reps=5):
    k = len(streams)
    mats = [hip.alloc_matrix(n, n, dev) for _ in range(k)]
    best = 1e9
    for _ in range(reps):
        for a in mats:
            a.copy_(K)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a, s in zip(mats, streams):
            with torch.cuda.stream(s):
                hip.potrf_(a, lookahead=False)
        torch.cuda.synchronize()
        best = min(best, 1e3 * (time.perf_counter() - t0))
    return best


print(f"n = {n}")
print("1 plain stream:", round(run([torch.cuda.Stream(device=dev)]), 3), "ms")
for k in (2, 3, 4):
    plain = run([torch.cuda.Stream(device=dev) for _ in range(k)])
    per = 256 // k
    masked = run([masked_stream(i * per, (i + 1) * per) for i in range(k)])
    # interleaved masks: CU c belongs to stream c % k (every stream keeps CUs on every XCD / shader engine)
    words = 8
    inter = []
    for i in range(k):
        mask = (ctypes.c_uint32 * words)()
        for cu in range(256):
            if cu % k == i:
                mask[cu // 32] |= 1 << (cu % 32)
        s = ctypes.c_void_p()
        assert hiprt.hipExtStreamCreateWithCUMask(ctypes.byref(s), words, mask) == 0
        inter.append(torch.cuda.ExternalStream(s.value, device=dev))
    print(f"{k} at once: plain streams {plain:.3f} ms, contiguous CU masks {masked:.3f} ms, interleaved CU masks {run(inter):.3f} ms")
