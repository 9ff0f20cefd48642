"""Host-side cost of one sharded_logpdf step (development aid): tiny n so that the GPU time is negligible, cProfile over 300 steps."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import c3_regressor, synthetic
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.parallel import sharded_logpdf
from gpar_amd.regression import _construct_gpar
p = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
eng = HipEngine(seed=1); set_engine(eng)
x_np, y_np = synthetic(n, 4, p)
reg = c3_regressor(); x = eng.tensor(x_np); y = eng.tensor(y_np); w = torch.ones_like(y)
gpar = _construct_gpar(reg, reg.vs, 4, p)
for _ in range(20): sharded_logpdf(gpar, x, y, w)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300): sharded_logpdf(gpar, x, y, w)
torch.cuda.synchronize(); print(f"p={p} n={n}: {1e3 * (time.perf_counter() - t0) / 300:.3f} ms per step")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): sharded_logpdf(gpar, x, y, w)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
