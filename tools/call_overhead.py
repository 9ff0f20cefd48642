"""Dev aid: cost of a tiny launch on the legacy default (null) stream against a created stream, before and after other streams
exist in the process (the null stream synchronises implicitly with every blocking stream)."""
import contextlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpar_amd import hip, _lib

lib = _lib.load(); dev = torch.device("cuda:0")
a = hip.alloc_matrix(8, 8, dev)


def measure(label, stream=None, n=2000):
    torch.cuda.synchronize()
    with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
        sp = hip.stream_ptr(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            lib.gpar_fill(a.data_ptr(), 8, 8, a.stride(0), 0.0, sp)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"{label}: enqueue {1e6 * (t1 - t0) / n:.2f} us/call, until drained {1e6 * (t2 - t0) / n:.2f} us/call")


measure("null stream, no other stream yet")
measure("null stream, again")
measure("null stream, third time", n=20000)
s = torch.cuda.Stream()
measure("null stream, one torch pool stream exists (idle)")
measure("the torch pool stream", s)
K = hip.alloc_matrix(8192, 8192, dev); K.zero_(); K.diagonal().fill_(1.0)
hip.potrf_(K)   # creates the library's non-blocking side stream
measure("null stream, after a look-ahead potrf (library side stream exists)")
s2 = torch.cuda.Stream(priority=-1)
measure("null stream, a second (high-priority) torch stream exists")
