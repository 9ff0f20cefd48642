"""Dev aid: C3 condition + predict (what bench.py's predict leg runs), for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
import bench
from gpar_amd.engine import HipEngine, set_engine

eng = HipEngine(device="cuda:0", seed=1); set_engine(eng)
x, y = bench.synthetic(16384, 4, 8)
reg = bench.c3_regressor()
reg.condition(x, y)
xs = np.random.default_rng(7).uniform(0, 1, (1024, 4))
reg.predict(xs, num_samples=2)
torch.cuda.synchronize(); t0 = time.perf_counter()
reg.predict(xs, num_samples=8)
torch.cuda.synchronize(); print("predict ms", 1e3 * (time.perf_counter() - t0))
