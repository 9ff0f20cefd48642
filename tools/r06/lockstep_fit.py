"""fit(iters=20), four layers: the lock-step rendezvous on / off (python tools/r06/lockstep_fit.py [n ...])"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import synthetic
from gpar_amd import optimise
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor
eng = HipEngine(seed=1); set_engine(eng)
kw = dict(scale=0.5, linear=True, nonlinear=False, noise=0.1, normalise_y=False)
for n in [int(a) for a in sys.argv[1:]] or [1024, 1536, 2048, 3072, 4096]:
    x, y = synthetic(n, 2, 4)
    for rows in ("0", "1"):
        os.environ["GPAR_FIT_LOCKSTEP_ROWS"] = rows
        times = []
        for rep in range(4):
            reg = GPARRegressor(**kw)
            before = optimise.evaluation_count()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            reg.fit(x, y, iters=20)
            torch.cuda.synchronize(); times.append(1e3 * (time.perf_counter() - t0))
            ev = optimise.evaluation_count() - before
        print(f"n={n} lockstep={'on' if rows == '1' else 'off'}: best {min(times[1:]):.1f} ms all {[round(t,1) for t in times]} evaluations {ev} rounds {getattr(reg, '_lockstep_rounds', None)}", flush=True)
