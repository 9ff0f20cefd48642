"""Host-side profile of small-problem calls (n = 400, p = 3): logpdf and predict; python tools/host_profile_small.py"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import synthetic
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor

eng = HipEngine(seed=3)
set_engine(eng)
n, p, ns, S = 400, 3, 200, 50
x, y = synthetic(n, 2, p)
xs = np.random.default_rng(1).uniform(0, 1, (ns, 2))
reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1, normalise_y=False)
reg.condition(x, y)
xd, yd = eng.tensor(x), eng.tensor(y)
for what, fn, reps in (("logpdf", lambda: reg.logpdf(xd, yd), 200), ("predict", lambda: reg.predict(xs, num_samples=S), 30)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(f"{what}: {1e3 * (time.perf_counter() - t0) / reps:.3f} ms per call")
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(reps):
        fn()
    prof.disable()
    torch.cuda.synchronize()
    pstats.Stats(prof).sort_stats("tottime").print_stats(22)
