"""Lock-step evaluations repeated: every repetition must return the same bits (python tools/check_lockstep_determinism.py [reps])."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bench import synthetic
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor

eng = HipEngine(seed=1)
set_engine(eng)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bad = 0
for n, p in [(4096, 4), (2100, 6), (8192, 8), (640, 5), (16384, 8), (8192, 16)]:
    x, y = synthetic(n, 2, p)
    xd, yd = eng.tensor(x), eng.tensor(y)
    reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1, normalise_y=False)
    vals = [float(reg.logpdf(xd, yd)) for _ in range(reps if n < 8192 else max(10, reps // 4))]
    distinct = sorted(set(vals))
    print(f"n={n} p={p}: {len(vals)} evaluations, {len(distinct)} distinct value(s): {distinct[:3]}")
    bad += len(distinct) > 1
sys.exit(1 if bad else 0)
