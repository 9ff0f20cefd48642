"""Host time of one gpar_logpdf_dense_grad call (enqueue only) and of a whole prepared-objective evaluation: python tools/r06/call_cost.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gpar_amd import fastfit
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests.test_fastfit import _data, _layer_objectives
eng = HipEngine(seed=1); set_engine(eng)
for n in [int(a) for a in sys.argv[1:]] or [25, 100, 400, 1024]:
    x, y, w = _data(n=n, m=2, p=3, seed=n)
    reg = GPARRegressor(scale=0.5, linear=True, nonlinear=False, noise=0.1, normalise_y=False)
    reg.condition(x, y, w)
    fast, fg, x0 = _layer_objectives(reg, eng, 2, None)
    for _ in range(20): fast.fg(x0)
    torch.cuda.synchronize()
    # whole evaluation
    t0 = time.perf_counter()
    for _ in range(200): fast.fg(x0)
    whole = (time.perf_counter() - t0) / 200
    # enqueue only: the library call without the synchronisation (queue runs ahead)
    ck = fastfit.compile_kernel(fast.kernel, fast.width)
    import ctypes
    p = fast._ptrs
    stream = torch.cuda.current_stream()
    def call():
        return fast.lib.gpar_logpdf_dense_grad(ctypes.byref(ck.fspec), ctypes.byref(ck.kspec), p["x"], fast.n, p["ldx"], p["y"], p["incy"], p["noise"], 1e-12,
            p["z"], p["zd"], p["ldz"], p["A"], p["lda"], p["X"], p["ldxw"], p["W"], p["ldw"], p["alpha"], p["work"], fast.nblocks, p["out"], p["half"], p["info"], 0, stream.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): call()
    enq = (time.perf_counter() - t0) / 50
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): call()
    torch.cuda.synchronize()
    gpu = (time.perf_counter() - t0) / 50
    t0 = time.perf_counter()
    for _ in range(200):
        fast._write_values(x0); fastfit.compile_kernel(fast.kernel, fast.width)
    prep = (time.perf_counter() - t0) / 200
    print(f"n={n}: whole evaluation {1e6 * whole:.0f} us; library call (enqueue, queue running ahead) {1e6 * enq:.0f} us; back-to-back calls incl. GPU {1e6 * gpu:.0f} us each; host prep {1e6 * prep:.0f} us", flush=True)
