"""BASELINE config C4 (n=65536, m=8, p=4, M=1024 inducing points): bound, one objective+gradient, a short fit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synthetic
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor
eng = HipEngine(seed=1); set_engine(eng)
n, m, p, M = 65536, 8, 4, 1024
x, y = synthetic(n, m, p)
reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1, normalise_y=False, x_ind=np.random.default_rng(3).uniform(0, 1, (M, m)))
def tic(): torch.cuda.synchronize(); return time.perf_counter()
xd, yd = eng.tensor(x), eng.tensor(y)
reg.logpdf(xd[:4096], yd[:4096])
t0 = tic(); v0 = float(reg.logpdf(xd, yd)); t1 = tic()
reg.vs.requires_grad(True)
t2 = tic(); val = reg.logpdf(xd, yd); val.backward(); t3 = tic()
reg.vs.requires_grad(False); reg.vs.requires_grad(True)
t2b = tic(); val = reg.logpdf(xd, yd); val.backward(); t3b = tic()   # (the first one pays for the allocator's first big blocks)
reg.vs.requires_grad(False)
t4 = tic(); reg.fit(x, y, iters=2); t5 = tic()
v1 = float(reg.logpdf(xd, yd))
print(f"C4 sparse: bound {v0:.3f} in {1e3*(t1-t0):.1f} ms; bound + gradient (all 4 layers) {1e3*(t3-t2):.1f} ms cold, {1e3*(t3b-t2b):.1f} ms warm; fit(iters=2) {t5-t4:.2f} s; bound after {v1:.3f}")
