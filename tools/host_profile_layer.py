"""Host-side self-time of the functions behind one pipelined evaluation (sorted by time spent IN each function):
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
prof = cProfile.Profile(); prof.enable()
for _ in range(20):
    reg.logpdf(x, y)
prof.disable(); torch.cuda.synchronize()
st = pstats.Stats(prof)
st.sort_stats("tottime").print_stats(32)
