"""Dev aid: repeatability of the pipelined C3 log marginal likelihood (bitwise)."""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bench import synthetic as _data
from gpar_amd.regression import GPARRegressor
n, m, p = 16384, 4, 8
x, y = _data(n, m, p)
reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, markov=2, noise=0.1, normalise_y=False)
vals = [float(reg.logpdf(x, y)) for _ in range(8)]
print(os.environ.get("GPAR_LAYER_PIPELINE"), os.environ.get("GPAR_POTRF_LOOKAHEAD"), [repr(v) for v in vals])
