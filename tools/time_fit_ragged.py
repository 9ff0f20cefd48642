import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bench import synthetic
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor
from gpar_amd import optimise
set_engine(HipEngine(seed=3))
for n in [int(a) for a in sys.argv[1:]] or (1000, 1500, 3000, 5000):
    x, y = synthetic(n, 2, 3)
    reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1)
    reg.fit(x, y, iters=3)
    torch.cuda.synchronize(); e0 = optimise.evaluation_count(); t0 = time.perf_counter()
    reg.fit(x, y, iters=10)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"n={n}: fit(10) {dt:.3f} s, {1e3 * dt / (optimise.evaluation_count() - e0):.2f} ms per evaluation", flush=True)
