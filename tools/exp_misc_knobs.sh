#!/bin/bash
# half-tile threshold and fit threads, same session
cd "$(dirname "$0")/.."
run() { python tools/run_config.py $CFG --evals 6 --warmup 2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); ms = sorted(d['ms']); print('$CFG $1', 'median', round(ms[len(ms)//2], 2))"; }
for CFG in C3 C5; do
  run default
  GPAR_GEMM_HALF_TILES=384 run half384
  GPAR_GEMM_HALF_TILES=512 run half512
  GPAR_GEMM_HALF_TILES=128 run half128
done
for th in 2 3 4; do
  GPAR_FIT_THREADS=$th python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bench import synthetic
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor
eng = HipEngine(seed=1); set_engine(eng)
x, y = synthetic(16384, 4, 8)
reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, markov=2, noise=0.1, normalise_y=True)
reg.fit(x, y, iters=1)
torch.cuda.synchronize(); t0 = time.perf_counter()
reg.fit(x, y, iters=4)
torch.cuda.synchronize(); print("fit threads", os.environ["GPAR_FIT_THREADS"], "fit(iters=4):", round(time.perf_counter() - t0, 3), "s")
PY
done
