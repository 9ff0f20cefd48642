"""Synthetic stand-ins for the reference's other example workloads (examples/paper/air_temp.py, eeg.py, exchange.py):
same model keywords, same missing-data patterns, same calls (fit, then predict with credible bounds) - on generated
data of the same shape, since the datasets are downloaded at run time there.  Each returns the standardised mean squared
error of the predictive means on the held-out entries (the metric those scripts print) for GPAR and for independent
GPs (markov=0).

    python tools/paper_workloads.py [air_temp|eeg|exchange] [--engine oracle] [--n 300]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def _smooth(rng, x, k=6, scale=1.0):
    """A random smooth function of x: a few sinusoids with random phases."""
    freq = rng.uniform(1.0, 8.0, k) * scale
    phase = rng.uniform(0, 2 * np.pi, k)
    amp = rng.standard_normal(k) / np.arange(1, k + 1)
    return np.sum(amp[None, :] * np.sin(freq[None, :] * x[:, None] + phase[None, :]), axis=1)


def air_temp(n=300, n_ind=31, seed=0):
    """air_temp.py:27-46: two stations, the second a noisy lagged function of the first; scattered missing values in both;
    inducing points on an even grid; replace + impute; lab's epsilon 1e-6."""
    rng = np.random.default_rng(seed)
    x = np.linspace(0.0, 10.0, n)
    f1 = _smooth(rng, x)
    f2 = 0.8 * np.interp(x - 0.15, x, f1) + 0.3 * _smooth(rng, x, scale=0.5)
    y = np.stack([f1, f2], axis=1) + 0.05 * rng.standard_normal((n, 2))
    train = y.copy()
    held = np.zeros_like(y, dtype=bool)
    held[int(0.55 * n) : int(0.65 * n), 1] = True  # a gap in the second station, as in the paper's figure
    held |= rng.random(y.shape) < 0.05
    train[held] = np.nan
    kw = dict(scale=0.2, linear=True, linear_scale=10.0, nonlinear=True, nonlinear_scale=1.0, noise=0.1, impute=True, replace=True,
              normalise_y=True, x_ind=np.linspace(x.min(), x.max(), n_ind))
    return x, train, y, held, kw, dict(epsilon=1e-6, latent=False)


def eeg(n=256, p=7, seed=1):
    """eeg.py:21-33: seven electrodes, the last three unobserved on the last 100 of 256 samples; nonlinear only."""
    rng = np.random.default_rng(seed)
    x = np.linspace(0.0, 1.0, n)
    base = _smooth(rng, x, k=10, scale=4.0)
    y = np.stack([base * rng.uniform(0.6, 1.2) + 0.4 * _smooth(rng, x, k=10, scale=4.0) for _ in range(p)], axis=1)
    y += 0.02 * rng.standard_normal(y.shape)
    train = y.copy()
    held = np.zeros_like(y, dtype=bool)
    held[n - int(100 * n / 256) :, p - 3 :] = True
    train[held] = np.nan
    kw = dict(scale=0.02, linear=False, nonlinear=True, nonlinear_scale=1.0, noise=0.01, impute=True, replace=False, normalise_y=True)
    return x, train, y, held, kw, dict(epsilon=1e-12, latent=True)


def exchange(n=251, p=6, seed=2):
    """exchange.py:21-35: correlated rate series, three of them with one missing block each; RQ kernels."""
    rng = np.random.default_rng(seed)
    x = np.linspace(0.0, 1.0, n)
    common = np.cumsum(rng.standard_normal(n)) / np.sqrt(n)
    y = np.stack([rng.uniform(0.5, 1.5) * common + 0.3 * np.cumsum(rng.standard_normal(n)) / np.sqrt(n) for _ in range(p)], axis=1)
    train = y.copy()
    held = np.zeros_like(y, dtype=bool)
    for col, (a, b) in zip([p - 3, p - 2, p - 1], [(0.2, 0.4), (0.45, 0.65), (0.7, 0.9)]):
        held[int(a * n) : int(b * n), col] = True
    train[held] = np.nan
    kw = dict(scale=0.1, linear=True, linear_scale=10.0, nonlinear=True, nonlinear_scale=1.0, rq=True, noise=0.01, impute=True,
              replace=False, normalise_y=True)
    return x, train, y, held, kw, dict(epsilon=1e-12, latent=False)


WORKLOADS = {"air_temp": air_temp, "eeg": eeg, "exchange": exchange}


def run(name, iters=30, num_samples=50, **size):
    from gpar_amd import GPARRegressor
    from gpar_amd.engine import get_engine

    x, train, truth, held, kw, opts = WORKLOADS[name](**size)
    eng = get_engine()
    previous, eng.epsilon = eng.epsilon, opts["epsilon"]
    out = {}
    try:
        for label, extra in [("gpar", {}), ("independent", dict(markov=0))]:
            model = GPARRegressor(**dict(kw, **extra))
            model.fit(x, train, iters=iters)
            mean, lower, upper = model.predict(x, num_samples=num_samples, credible_bounds=True, latent=opts["latent"])
            err = (mean - truth) ** 2
            cols = held.any(axis=0)
            var = np.array([np.mean((truth[held[:, j], j] - np.nanmean(train[:, j])) ** 2) if cols[j] else np.nan for j in range(truth.shape[1])])
            smse = np.array([np.mean(err[held[:, j], j]) / var[j] if cols[j] else np.nan for j in range(truth.shape[1])])
            out[label] = {"smse": smse, "finite": bool(np.all(np.isfinite(mean)) and np.all(lower <= upper))}
    finally:
        eng.epsilon = previous
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="all", choices=["all"] + sorted(WORKLOADS))
    ap.add_argument("--engine", choices=["hip", "oracle"], default="hip")
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    from gpar_amd.engine import HipEngine, set_engine

    if args.engine == "oracle":
        from oracle.engine import OracleEngine

        set_engine(OracleEngine(seed=1))
    else:
        set_engine(HipEngine(seed=1))
    for name in sorted(WORKLOADS) if args.workload == "all" else [args.workload]:
        res = run(name, iters=args.iters, **({"n": args.n} if args.n else {}))
        for label, r in res.items():
            print(f"{name:9s} {label:12s} SMSE on the held-out entries per output {np.round(r['smse'], 3)}")
