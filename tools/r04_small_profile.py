"""Where a small lock-step evaluation (n = 512, p = 4) spends its time: wall-clock per call, host profile, and - under rocprofv3 -
the kernel timeline.   python tools/r04_small_profile.py [n] [p]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
p = int(sys.argv[2]) if len(sys.argv) > 2 else 4
eng = HipEngine(seed=3)
set_engine(eng)
x, y = synthetic(n, 2, p)
reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1, normalise_y=False)
xd, yd = eng.tensor(x), eng.tensor(y)
fn = lambda: float(reg.logpdf(xd, yd))
for _ in range(5):
    fn()
for reps in (200, 200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(f"n={n} p={p}: {1e3 * (time.perf_counter() - t0) / reps:.3f} ms per call", flush=True)
if os.environ.get("PROFILE_HOST", "1") == "1":
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(200):
        fn()
    prof.disable()
    torch.cuda.synchronize()
    pstats.Stats(prof).sort_stats("tottime").print_stats(28)
