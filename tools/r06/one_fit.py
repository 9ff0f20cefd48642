"""One configuration of tools/r06/small_fit.py for a profiler: python tools/r06/one_fit.py n threads [fast] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import synthetic
from gpar_amd import optimise
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor

n, threads = int(sys.argv[1]), sys.argv[2]
fast = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
if threads != "default":
    os.environ["GPAR_FIT_THREADS"] = threads
eng = HipEngine(seed=1)
set_engine(eng)
x, y = synthetic(n, 2, 4)
kw = dict(scale=0.5, linear=True, nonlinear=False, noise=0.1, normalise_y=False)
for rep in range(reps):
    reg = GPARRegressor(**kw)
    reg.fast_fit = fast
    before = optimise.evaluation_count()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reg.fit(x, y, iters=20)
    torch.cuda.synchronize()
    dt = 1e3 * (time.perf_counter() - t0)
    print(f"n={n} threads={threads} fast={int(fast)}: {dt:.1f} ms, {optimise.evaluation_count() - before} evaluations", flush=True)
