import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
eng = HipEngine(seed=1); set_engine(eng)
rng = np.random.default_rng(5)
n, m, p, S, ns = 900, 2, 4, 7, 300
x = rng.uniform(0, 1, (n, m)); cols = []
for i in range(p):
    base = np.sin(2*np.pi*(x @ rng.uniform(0.5,1.5,m)) + i)
    if cols: base = base + 0.5*cols[-1]**2
    cols.append(base + 0.1*rng.standard_normal(n))
y = np.stack(cols, 1); y = (y - y.mean(0))/y.std(0)
xs = np.random.default_rng(6).uniform(0,1,(ns,m)); w = np.random.default_rng(7).uniform(0.5,2.0,(ns,p))
for kw, latent in [(dict(), True), (dict(), False), (dict(markov=1), True), (dict(input_linear=True, rq=True), True), (dict(linear_scale=3.0, scale=0.3), False)]:
    out = {}
    for mode in "10":
        os.environ["GPAR_LINEAR_TAIL"] = mode
        reg = GPARRegressor(**dict(dict(scale=0.5, linear=True, nonlinear=False, noise=0.1), **kw)); reg.condition(x, y); eng.seed(33)
        out[mode] = np.stack(reg.sample(xs, w=w, posterior=True, num_samples=S, latent=latent))
    d = np.abs(out["1"] - out["0"])
    print(kw, latent, "max abs", d.max(), "per layer", d.max(axis=(0,1)), "scale", np.abs(out["0"]).max(axis=(0,1)))
