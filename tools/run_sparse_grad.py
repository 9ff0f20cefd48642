"""C4 (n=65536, m=8, p=4, M=1024): the bound + its gradient, alone, for profiling:  python tools/run_sparse_grad.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synthetic
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor
eng = HipEngine(seed=1); set_engine(eng)
n, m, p, M = 65536, 8, 4, 1024
x, y = synthetic(n, m, p)
reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1, normalise_y=False, x_ind=np.random.default_rng(3).uniform(0, 1, (M, m)))
xd, yd = eng.tensor(x), eng.tensor(y)
reg.logpdf(xd[:4096], yd[:4096])
reg.vs.requires_grad(True)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    time.sleep(0.05)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    val = reg.logpdf(xd, yd); val.backward()
    torch.cuda.synchronize(); print(f"bound + gradient: {1e3 * (time.perf_counter() - t0):.1f} ms")
if len(sys.argv) > 2 and sys.argv[2] == "profile":
    import cProfile, pstats
    prof = cProfile.Profile(); prof.enable()
    val = reg.logpdf(xd, yd); val.backward(); torch.cuda.synchronize()
    prof.disable()
    pstats.Stats(prof).sort_stats("tottime").print_stats(22)
