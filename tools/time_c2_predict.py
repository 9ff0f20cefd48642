"""predict(num_samples=100, n* = 2048) of the C2 model (n = 4096, m = 2, p = 4, the reference's default output dependence:
linear=True, nonlinear=False) with and without the shared-solve routine for linear output parts (GPAR_LINEAR_TAIL)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import synthetic
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor

eng = HipEngine(seed=1)
set_engine(eng)
n, m, p, S, ns = 4096, 2, 4, 100, 2048
if len(sys.argv) > 1:
    n, p, S, ns = (int(v) for v in sys.argv[1:5])
x, y = synthetic(n, m, p)
xs = np.random.default_rng(5).uniform(0, 1, (ns, m))
for mode in ("1", "0", "1", "0"):
    os.environ["GPAR_LINEAR_TAIL"] = mode
    reg = GPARRegressor(scale=0.5, linear=True, nonlinear=False, noise=0.1)
    reg.condition(x, y)
    eng.seed(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mean = reg.predict(xs, num_samples=S, latent=True)
    torch.cuda.synchronize()
    print(f"n={n} p={p} S={S} n*={ns} GPAR_LINEAR_TAIL={mode}: predict {1e3 * (time.perf_counter() - t0):.1f} ms  (mean |.| {np.abs(mean).mean():.9f})", flush=True)
