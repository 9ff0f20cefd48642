import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synthetic
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor
eng = HipEngine(seed=1); set_engine(eng)
x, y = synthetic(65536, 8, 4)
reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1, normalise_y=False, x_ind=np.random.default_rng(3).uniform(0, 1, (1024, 8)))
xd, yd = eng.tensor(x), eng.tensor(y)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); v = float(reg.logpdf(xd, yd)); torch.cuda.synchronize(); print("C4 logpdf ms", 1e3 * (time.perf_counter() - t0))
