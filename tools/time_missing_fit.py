import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bench import synthetic
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor
from gpar_amd import optimise
set_engine(HipEngine(seed=3))
n, p = 2000, 4
x, y = synthetic(n, 2, p)
y[np.random.default_rng(0).random(y.shape) < 0.1] = np.nan
for name, kw in (("impute", dict(impute=True)), ("no-impute", dict(impute=False))):
    reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1, **kw)
    reg.fit(x, y, iters=3)
    torch.cuda.synchronize(); e0 = optimise.evaluation_count(); t0 = time.perf_counter()
    reg.fit(x, y, iters=10)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: fit(10) n={n} p={p} 10% missing: {dt:.3f} s, {1e3 * dt / (optimise.evaluation_count() - e0):.2f} ms per evaluation, logpdf {float(reg.logpdf(x, y)):.6f}", flush=True)
