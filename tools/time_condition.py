import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from bench import synthetic
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor
eng = HipEngine(seed=3); set_engine(eng)
for n, p in [(400, 3), (2048, 4), (4096, 4), (8192, 8)]:
    x, y = synthetic(n, 2, p)
    reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1, normalise_y=False)
    ts = []
    for i in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); reg.condition(x, y); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    xd, yd = eng.tensor(x), eng.tensor(y)
    tl = []
    for i in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); reg.logpdf(xd, yd); torch.cuda.synchronize(); tl.append(1e3 * (time.perf_counter() - t0))
    print(f"n={n} p={p}: condition {min(ts[1:]):.2f} ms, logpdf {min(tl[1:]):.2f} ms")
