import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bench import synthetic
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor
from gpar_amd import optimise
set_engine(HipEngine(seed=3))
n, M = 20000, 300
x, y = synthetic(n, 2, 3)
reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1, x_ind=np.random.default_rng(2).uniform(0, 1, (M, 2)))
reg.fit(x, y, iters=2)
for rep in range(2):
    torch.cuda.synchronize(); e0 = optimise.evaluation_count(); t0 = time.perf_counter()
    reg.fit(x, y, iters=5)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"sparse fit(5) n={n} M={M}: {dt:.3f} s, {optimise.evaluation_count() - e0} evaluations, {1e3 * dt / (optimise.evaluation_count() - e0):.2f} ms each", flush=True)
