"""Small problems (the reference's own examples are n ~ 100-1000): logpdf and fit(iters) wall time per size on the HIP engine,
and the same calls on the CPU oracle engine for scale.   python tools/time_small_sizes.py [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import synthetic
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor
from oracle.engine import OracleEngine

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
hip = HipEngine(seed=3)
rows = {}
# (all HIP sizes first: the CPU engine's BLAS worker threads keep spinning for a while after a call and slow the host side of
# the - host-bound - small HIP calls that follow: n = 512 measured 2.2 ms behind a CPU run, 0.58 ms otherwise)
for name, eng in (("hip", hip), ("cpu-oracle", OracleEngine(seed=3))):
    for n in (128, 512, 1024, 2048):
        if name == "cpu-oracle" and n > 1024:
            continue
        x, y = synthetic(n, 2, 3)
        row = rows.setdefault(n, {})
        set_engine(eng)
        reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, noise=0.1, normalise_y=True)
        reg.condition(x, y)
        for _ in range(3):
            reg.logpdf(x, y)
        t0 = time.perf_counter()
        reps = 50 if name == "hip" else 5
        for _ in range(reps):
            reg.logpdf(x, y)
        if name == "hip":
            torch.cuda.synchronize()
        row[name + " logpdf ms"] = 1e3 * (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        reg.fit(x, y, iters=iters)
        if name == "hip":
            torch.cuda.synchronize()
        row[name + f" fit({iters}) s"] = time.perf_counter() - t0
for n, row in rows.items():
    print(n, {k: round(v, 3) for k, v in row.items()}, flush=True)
