"""fit(iters=20) at C3 (n = 16384, m = 4, p = 8) in this process: wall-clock, evaluations, compiled-kernel statistics (development aid)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import c3_regressor, synthetic
from gpar_amd import _lib, optimise
from gpar_amd.engine import HipEngine, set_engine
eng = HipEngine(seed=1); set_engine(eng)
x, y = synthetic(16384, 4, 8)
reg = c3_regressor()
torch.cuda.synchronize(); t0 = time.perf_counter()
reg.fit(x, y, iters=int(sys.argv[1]) if len(sys.argv) > 1 else 20)
torch.cuda.synchronize(); t1 = time.perf_counter()
cs = [ctypes.c_int(), ctypes.c_int(), ctypes.c_int()]
_lib.load().gpar_jit_stats(*[ctypes.byref(c) for c in cs])
print(f"fit: {t1 - t0:.3f} s, {optimise.evaluation_count()} evaluations, jit compiled/failed/cached {[c.value for c in cs]}, GPAR_JIT_PREPARE={os.environ.get('GPAR_JIT_PREPARE', '1')}")
