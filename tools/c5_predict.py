"""BASELINE config C5 end to end: n=8192, m=3, p=16, per + rq kernels, predict with num_samples=200 at n*=2048."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synthetic
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor
eng = HipEngine(seed=1); set_engine(eng)
n, m, p = 8192, 3, 16
S = int(sys.argv[1]) if len(sys.argv) > 1 else 200
x, y = synthetic(n, m, p)
reg = GPARRegressor(scale=0.5, per=True, rq=True, linear=True, nonlinear=True, noise=0.1)
def tic(): torch.cuda.synchronize(); return time.perf_counter()
t0 = tic(); reg.condition(x, y); v = float(reg.logpdf(x, y)); t1 = tic()
xs = np.random.default_rng(2).uniform(0, 1, (2048, m))
mean, lo, hi = reg.predict(xs, num_samples=S, credible_bounds=True); t2 = tic()
print(f"C5: logpdf {v:.6f} ({1e3*(t1-t0):.0f} ms incl. first-call setup); predict S={S} n*=2048: {t2-t1:.2f} s; "
      f"finite={np.isfinite(mean).all() and np.isfinite(lo).all() and np.isfinite(hi).all()}, bounds ordered={(lo <= hi).all()}, "
      f"peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
t3 = tic(); mean2, lo2, hi2 = reg.predict(xs, num_samples=S, credible_bounds=True); t4 = tic()
mm, lm, hm = reg.predict(xs, num_samples=S, credible_bounds=True, marginal=True); t5 = tic()
print(f"    again: {t4-t3:.2f} s;   marginal=True: {t5-t4:.2f} s;   |mean - marginal mean| max {np.abs(mean2 - mm).max():.3f} "
      f"(bounds half-width median {np.median(hi2 - lo2) / 2:.3f})")
