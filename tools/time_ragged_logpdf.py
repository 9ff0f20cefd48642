import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bench import synthetic
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor
set_engine(HipEngine(seed=3))
for n in (960, 1000, 1024, 2000, 2048, 3000, 3072):
    x, y = synthetic(n, 2, 8)
    reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1)
    xd, yd = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
    for _ in range(5): reg.logpdf(xd, yd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): reg.logpdf(xd, yd)
    torch.cuda.synchronize(); print(f"n={n} p=8 logpdf: {1e3 * (time.perf_counter() - t0) / 30:.3f} ms", flush=True)
