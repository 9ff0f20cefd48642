"""Dev aid: one short C3 fit (what bench.py's fit leg runs), for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch
import bench
from gpar_amd.engine import HipEngine, set_engine

eng = HipEngine(device="cuda:0", seed=1); set_engine(eng)
x, y = bench.synthetic(16384, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 4)
reg = bench.c3_regressor()
torch.cuda.synchronize(); t0 = time.perf_counter()
reg.fit(x, y, iters=2)
torch.cuda.synchronize(); print("fit ms", 1e3 * (time.perf_counter() - t0))
