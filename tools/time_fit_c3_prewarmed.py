"""fit(iters=20) at C3 with every run-time compiled kernel already in the process's cache: what first-use compilation costs inside a fit (development aid)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import c3_regressor, synthetic
from gpar_amd import _lib, optimise
from gpar_amd.engine import HipEngine, set_engine
eng = HipEngine(seed=1); set_engine(eng)
x, y = synthetic(16384, 4, 8)
reg = c3_regressor()
t0 = time.perf_counter()
reg._prepare_kernels(4, 8, 16384, training=True)
t1 = time.perf_counter()
print(f"compiling every structure (concurrently): {t1 - t0:.2f} s")
for rep in range(2):
    reg = c3_regressor()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reg.fit(x, y, iters=20)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"fit #{rep}: {t1 - t0:.3f} s, {optimise.evaluation_count()} evaluations so far")
