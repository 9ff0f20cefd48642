"""BASELINE.json config grid on one MI355X: wall-clock of logpdf (and condition + a few posterior samples) per config.
Development/measurement aid; results are copied into DESIGN.md."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synthetic
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor

eng = HipEngine(seed=1); set_engine(eng)
def tic(): torch.cuda.synchronize(); return time.perf_counter()
CONFIGS = {
    "C2": dict(n=4096, m=2, p=4, kw=dict(scale=0.5, linear=True, nonlinear=False, noise=0.1)),
    "C3": dict(n=16384, m=4, p=8, kw=dict(scale=0.5, linear=True, nonlinear=True, markov=2, noise=0.1)),
    "C4": dict(n=65536, m=8, p=4, kw=dict(scale=0.5, linear=True, nonlinear=True, noise=0.1), M=1024),
    "C5": dict(n=8192, m=3, p=16, kw=dict(scale=0.5, per=True, rq=True, linear=True, nonlinear=True, noise=0.1)),
}
which = sys.argv[1:] or list(CONFIGS)
out = {}
for name in which:
    c = CONFIGS[name]
    x, y = synthetic(c["n"], c["m"], c["p"])
    kw = dict(c["kw"], normalise_y=False)
    if "M" in c:
        kw["x_ind"] = np.random.default_rng(3).uniform(0, 1, (c["M"], c["m"]))
    reg = GPARRegressor(**kw)
    reg.logpdf(x[:512], y[:512])
    xd, yd = eng.tensor(x), eng.tensor(y)
    t0 = tic(); v = float(reg.logpdf(xd, yd)); t1 = tic(); v2 = float(reg.logpdf(xd, yd)); t2 = tic()
    reg.condition(x, y)
    xs = np.random.default_rng(5).uniform(0, 1, (1024, c["m"]))
    t3 = tic(); s = reg.sample(xs, posterior=True, num_samples=2, latent=True); t4 = tic()
    out[name] = {"logpdf": v, "logpdf_ms_first": 1e3 * (t1 - t0), "logpdf_ms": 1e3 * (t2 - t1), "condition_plus_2_samples_ms": 1e3 * (t4 - t3),
                 "repeatable": v == v2, "sample_finite": bool(np.all(np.isfinite(np.stack(s))))}
    print(name, json.dumps(out[name]), flush=True)
    del reg, xd, yd; torch.cuda.empty_cache()
