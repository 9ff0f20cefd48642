"""C2's predict leg alone: python tools/r06/c2_predict.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench import synthetic, GRID
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor
eng = HipEngine(seed=1); set_engine(eng)
cfg = GRID["C2"]; n, m, p = cfg["n"], cfg["m"], cfg["p"]
x_np, y_np = synthetic(n, m, p)
reg = GPARRegressor(**dict(cfg["kw"], normalise_y=False))
reg.condition(x_np, y_np)
xs = np.random.default_rng(2).uniform(0, 1, (2048, m))
for S in (4, 100, 100, 100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reg.predict(xs, num_samples=S)
    torch.cuda.synchronize(); print(f"predict({S}) {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
