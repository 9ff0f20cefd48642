"""predict (joint ancestral sampling) at a few sizes: per-sample n* x n* blocks in lock-step against the stream version.
   python tools/time_predict_sizes.py [n:p:nstar:S ...]"""
import os
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
cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(400, 3, 200, 50), (2048, 4, 512, 100), (4096, 4, 1024, 50), (16384, 8, 2048, 20)]
for n, p, ns, S in cases:
    x, y = synthetic(n, 2, p)
    xs = np.random.default_rng(1).uniform(0, 1, (ns, 2))
    reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1, normalise_y=False)
    reg.condition(x, y)
    line = [f"n={n} p={p} n*={ns} S={S}:"]
    for name, env in (("streams", {"GPAR_LAYER_BATCH_ROWS": "0"}), ("lockstep", {})):
        os.environ.update(env)
        ts = []
        for i in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mean = reg.predict(xs, num_samples=S)
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        for k in env:
            os.environ.pop(k)
        line.append(f"{name} {min(ts[1:]):.1f} ms (mean|.| {np.abs(mean).mean():.6f})")
    print("  ".join(line), flush=True)
