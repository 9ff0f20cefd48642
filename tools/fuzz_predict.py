"""Exploratory differential runs of the sampling path at blocked sizes: reg.sample(xs, num_samples=2, posterior=True) with noisy
(latent=False) outputs on the HIP engine against the numpy engine on the shared Philox stream, and predict's reduction (development aid)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from tests.conftest import make_engine, to_np
from gpar_amd.engine import set_engine
from gpar_amd.regression import GPARRegressor

def case(seed):
    rng = np.random.default_rng(7000 + seed)
    n, ns = int(rng.integers(200, 2600)), int(rng.integers(100, 1600))
    m, p = int(rng.integers(1, 3)), int(rng.integers(1, 4))
    kw = dict(scale=float(rng.uniform(0.2, 1.0)), noise=float(10 ** rng.uniform(-2.5, -0.5)), normalise_y=bool(rng.integers(2)), linear=bool(rng.integers(2)),
              nonlinear=bool(rng.integers(2)), rq=bool(rng.integers(2)), per=bool(rng.integers(4) == 0), markov=[None, 1][int(rng.integers(2))],
              impute=bool(rng.integers(2)), replace=bool(rng.integers(3) == 0))
    if rng.integers(3) == 0:
        M = int(rng.integers(20, 300))
        kw["x_ind"] = np.linspace(0, 1, M)[:, None] if m == 1 else rng.uniform(0, 1, (M, m))
    x = rng.uniform(0, 1, (n, m))
    cols = []
    for i in range(p):
        base = np.sin(2 * np.pi * (x @ rng.uniform(0.5, 1.5, m)) + i)
        if cols:
            base = base + 0.4 * cols[-1]
        cols.append(base + 0.1 * rng.standard_normal(n))
    y = np.stack(cols, axis=1)
    if rng.integers(2):
        y[rng.random(y.shape) < 0.1] = np.nan
        y[0] = 0.2
    xs = rng.uniform(0, 1, (ns, m))
    return kw, x, y, xs

def run(kind, kw, x, y, xs, latent):
    eng = make_engine(kind, seed=11)
    prev = set_engine(eng)
    try:
        reg = GPARRegressor(**kw)
        reg.condition(x, y)
        smp = [to_np(s_) for s_ in reg.sample(xs, num_samples=2, latent=latent, posterior=True)]
        mean, lo, hi = reg.predict(xs, num_samples=4, latent=latent, credible_bounds=True)
        return np.stack(smp), np.stack([to_np(mean), to_np(lo), to_np(hi)])
    finally:
        set_engine(prev)

bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    kw, x, y, xs = case(seed)
    desc = {k: (v.shape if hasattr(v, "shape") else v) for k, v in kw.items()}
    t0 = time.time()
    try:
        latent = len(sys.argv) > 3 and sys.argv[3] == "latent"   # noise-free draws: covariances that are singular but for the jitter
        hs, hp = run("hip", kw, x, y, xs, latent)
        os_, op = run("oracle", kw, x, y, xs, latent)
        scale = max(1.0, np.abs(os_).max())
        ds, dp = np.max(np.abs(hs - os_)) / scale, np.max(np.abs(hp - op)) / scale
        tol = 1e-4 if "x_ind" in kw else 1e-7
        if latent:
            tol = 1e-2   # (draws of a numerically singular covariance are determined to ~sqrt(jitter) only)
        flag = "" if (ds <= tol and dp <= tol) else "  <<<<<< MISMATCH"
        bad += bool(flag)
        print(seed, x.shape, xs.shape[0], y.shape[1], "sparse" if "x_ind" in kw else "dense", "dsample %.1e dpredict %.1e  %.1fs%s" % (ds, dp, time.time() - t0, flag), desc if flag else "", flush=True)
    except Exception as e:
        bad += 1
        print(seed, "FAILED", type(e).__name__, str(e)[:200], x.shape, xs.shape, desc, flush=True)
print("bad:", bad)
