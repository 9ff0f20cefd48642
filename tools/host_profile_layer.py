"""Host-side self-time of the functions behind one evaluation (sorted by time spent IN each function), asynchronous GPU work:
    python tools/host_profile_layer.py C2"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpar_amd.engine import HipEngine, set_engine
from tools.run_config import build

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
eng = HipEngine(seed=1); set_engine(eng)
cfg, reg, x, y = build(name, eng)
for _ in range(3):
    reg.logpdf(x, y)
torch.cuda.synchronize()
os.environ["GPAR_LAYER_PIPELINE"] = "0"
# without the profiler: host time to ENQUEUE one evaluation (no sync inside)
from gpar_amd.regression import _construct_gpar, _default_weights
w = _default_weights(*y.shape)
t0 = time.perf_counter()
N = 20
for _ in range(N):
    gpar = _construct_gpar(reg, reg.vs, cfg["m"], cfg["p"])
    v = gpar.logpdf(x, y, w, only_last_layer=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"{name}: host enqueue {1e3 * (t1 - t0) / N:.3f} ms per evaluation = {1e3 * (t1 - t0) / N / cfg['p']:.3f} ms per layer (no pipelining)")
prof = cProfile.Profile(); prof.enable()
for _ in range(10):
    gpar = _construct_gpar(reg, reg.vs, cfg["m"], cfg["p"])
    v = gpar.logpdf(x, y, w, only_last_layer=False)
prof.disable(); torch.cuda.synchronize()
pstats.Stats(prof).sort_stats("tottime").print_stats(30)
