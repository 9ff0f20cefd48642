import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bench import synthetic
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor
set_engine(HipEngine(seed=3))
n, ns, S, M = 20000, 1000, 50, 300
x, y = synthetic(n, 2, 3)
xs = np.random.default_rng(1).uniform(0, 1, (ns, 2))
reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1, x_ind=np.random.default_rng(2).uniform(0, 1, (M, 2)))
t0 = time.perf_counter(); reg.condition(x, y); lp = reg.logpdf(x, y); torch.cuda.synchronize(); print(f"condition + logpdf: {1e3 * (time.perf_counter() - t0):.1f} ms")
for _ in range(2): reg.logpdf(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): reg.logpdf(x, y)
torch.cuda.synchronize(); print(f"sparse logpdf n={n} M={M}: {1e3 * (time.perf_counter() - t0) / 5:.2f} ms per call")
for _ in range(2): reg.predict(xs, num_samples=S)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): reg.predict(xs, num_samples=S)
torch.cuda.synchronize(); print(f"sparse predict n*={ns} S={S}: {1e3 * (time.perf_counter() - t0) / 5:.2f} ms per call")
t0 = time.perf_counter(); reg.fit(x, y, iters=5); torch.cuda.synchronize(); print(f"sparse fit(5): {time.perf_counter() - t0:.2f} s")
