import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synthetic, GRID
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor
eng = HipEngine(seed=1); set_engine(eng)
cfg = GRID["C2"]; x, y = synthetic(cfg["n"], cfg["m"], cfg["p"]); kw = dict(cfg["kw"], normalise_y=False)
for t in ("2", "4", "2", "4"):
    os.environ["GPAR_FIT_THREADS"] = t
    best = 1e9
    for rep in range(4):
        reg = GPARRegressor(**kw); torch.cuda.synchronize(); t0 = time.perf_counter(); reg.fit(x, y, iters=20); torch.cuda.synchronize()
        if rep: best = min(best, time.perf_counter() - t0)
    print(f"C2 fit threads={t}: {1e3*best:.1f} ms")
