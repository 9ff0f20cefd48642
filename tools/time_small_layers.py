"""Log marginal likelihood of p independent layers of n rows each: lock-step batch (with / without its look-ahead) against
layers on separate streams.   python tools/time_small_layers.py [n:p ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import synthetic
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor

eng = HipEngine()
set_engine(eng)
cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(512, 4), (1024, 4), (2048, 4), (2048, 8), (3072, 4), (4096, 4), (4096, 8), (6144, 4), (8192, 4)]
MODES = {"streams": {"GPAR_LAYER_BATCH_ROWS": "0"}, "lockstep": {"GPAR_LAYER_BATCH_ROWS": "100000", "GPAR_POTRF_BATCH_LOOKAHEAD": "0"},
         "lockstep+lookahead": {"GPAR_LAYER_BATCH_ROWS": "100000", "GPAR_POTRF_BATCH_LOOKAHEAD": "1"}}
for n, p in cases:
    x, y = synthetic(n, 2, p)
    reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1, normalise_y=False)
    xd, yd = eng.tensor(x), eng.tensor(y)
    line = [f"n={n} p={p}:"]
    for name, env in MODES.items():
        os.environ.update(env)
        ts = []
        for i in range(9):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            v = float(reg.logpdf(xd, yd))
            ts.append(1e3 * (time.perf_counter() - t0))
        for k in env:
            os.environ.pop(k)
        line.append(f"{name} {sorted(ts[2:])[len(ts[2:]) // 2]:.2f} ms ({v:.6f})")
    print("  ".join(line), flush=True)
