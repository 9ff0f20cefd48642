"""Dev aid: host-side profile of one-layer logpdf steps (what a rank of an 8-GPU run executes per step)."""
import cProfile, pstats, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.parallel import sharded_logpdf
from gpar_amd.regression import _construct_gpar

eng = HipEngine(device="cuda:0", seed=1); set_engine(eng)
n, m, p = 16384, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 1
x_np, y_np = bench.synthetic(n, m, p)
reg = bench.c3_regressor()
x, y = eng.tensor(x_np), eng.tensor(y_np); w = torch.ones_like(y)
gpar = _construct_gpar(reg, reg.vs, m, p)
for _ in range(3): sharded_logpdf(gpar, x, y, w)
torch.cuda.synchronize()
# time from step start until the last launch is enqueued vs until the value is back
t0 = time.perf_counter()
for _ in range(10): sharded_logpdf(gpar, x, y, w)
torch.cuda.synchronize(); print("ms/step", 1e2 * (time.perf_counter() - t0))
pr = cProfile.Profile(); pr.enable()
for _ in range(10): sharded_logpdf(gpar, x, y, w)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
