import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bench import synthetic
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor
set_engine(HipEngine(seed=3))
n, p = 2000, 4
x, y = synthetic(n, 2, p)
rng = np.random.default_rng(0)
y[rng.random(y.shape) < 0.1] = np.nan
for name, kw in (("impute", dict(impute=True)), ("no-impute", dict(impute=False)), ("replace", dict(impute=True, replace=True))):
    reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1, **kw)
    for _ in range(3): reg.logpdf(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): reg.logpdf(x, y)
    torch.cuda.synchronize(); print(f"{name}: logpdf n={n} p={p} 10% missing: {1e3 * (time.perf_counter() - t0) / 10:.2f} ms", flush=True)
