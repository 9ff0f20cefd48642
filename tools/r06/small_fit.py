"""fit(iters=20) in the small-problem regime (bench.py's `small_n` model: m = 2, p = 4, linear output dependence), prepared
objective (gpar_amd/fastfit.py) against the general route, 1 / 2 / 4 host threads:  python tools/r06/small_fit.py [n ...]
PROFILE=1: cProfile of one single-thread fit per size."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from bench import synthetic
from gpar_amd import optimise
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor

eng = HipEngine(seed=1)
set_engine(eng)
sizes = [int(a) for a in sys.argv[1:]] or [100, 400, 1024, 2048]
kw = dict(scale=0.5, linear=True, nonlinear=False, noise=0.1, normalise_y=False)
for n in sizes:
    x, y = synthetic(n, 2, 4)
    for fast in (True, False):
        for threads in ("1", "2", "4", None):
            if threads is None:
                os.environ.pop("GPAR_FIT_THREADS", None)
            else:
                os.environ["GPAR_FIT_THREADS"] = threads
            times = []
            for rep in range(4):
                reg = GPARRegressor(**kw)
                reg.fast_fit = fast
                before = optimise.evaluation_count()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                reg.fit(x, y, iters=20)
                torch.cuda.synchronize()
                times.append(1e3 * (time.perf_counter() - t0))
                evals = optimise.evaluation_count() - before
            print(f"n={n:5d} fast={int(fast)} threads={threads or 'default'}: fit(iters=20) best {min(times[1:]):7.1f} ms  all {[round(t, 1) for t in times]}  "
                  f"evaluations {evals}  -> {1e3 * min(times[1:]) / evals:.0f} us per evaluation", flush=True)
    if os.environ.get("PROFILE"):
        os.environ["GPAR_FIT_THREADS"] = "1"
        reg = GPARRegressor(**kw)
        prof = cProfile.Profile()
        prof.enable()
        reg.fit(x, y, iters=20)
        prof.disable()
        pstats.Stats(prof).sort_stats("tottime").print_stats(30)
