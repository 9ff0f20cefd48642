"""BASELINE.json configs[0]: the synthetic experiment of the GPAR paper (reference: examples/paper/synthetic.py) through
this package - three outputs that depend on one another, 25 noisy observations of each, GPAR against independent GPs
(markov=0).  Prints the RMSE of the latent predictive means against the true functions; GPAR should win clearly on the
dependent outputs.  `--engine oracle` runs the CPU oracle instead of the HIP library (test infrastructure only)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def dataset(seed=1, n=200, noise=0.1):
    x = np.linspace(0, 1, n)
    f1 = -np.sin(10 * np.pi * (x + 1)) / (2 * x + 1) - x**4
    f2 = np.cos(f1) ** 2 + np.sin(3 * x)
    f3 = f2 * f1**2 + 3 * x
    f = np.stack([f1, f2, f3], axis=1)
    y = f + noise * np.random.default_rng(seed).standard_normal(f.shape)
    return x, f, x[::8], y[::8]


def run(iters=200, num_samples=200, seed=1):
    from gpar_amd import GPARRegressor

    x, f, x_obs, y_obs = dataset(seed)
    common = dict(scale=0.1, linear=True, linear_scale=10.0, nonlinear=True, nonlinear_scale=0.1, noise=0.1, normalise_y=False)
    out = {}
    for name, kw in [("gpar", dict(impute=True, replace=False)), ("independent", dict(markov=0))]:
        model = GPARRegressor(**common, **kw)
        model.fit(x_obs, y_obs, iters=iters)
        mean, lower, upper = model.predict(x, num_samples=num_samples, credible_bounds=True, latent=True)
        out[name] = {"rmse": np.sqrt(np.mean((mean - f) ** 2, axis=0)), "coverage": np.mean((lower <= f) & (f <= upper), axis=0),
                     "logpdf": float(model.logpdf(x_obs, y_obs))}
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--engine", choices=["hip", "oracle"], default="hip")
    ap.add_argument("--iters", type=int, default=200)
    args = ap.parse_args()
    from gpar_amd.engine import set_engine

    if args.engine == "oracle":
        from oracle.engine import OracleEngine

        set_engine(OracleEngine(seed=1))
    else:
        from gpar_amd.engine import HipEngine

        set_engine(HipEngine(seed=1))
    for name, r in run(iters=args.iters).items():
        print(f"{name:12s} rmse per output {np.round(r['rmse'], 4)}  95% coverage {np.round(r['coverage'], 2)}  logpdf(train) {r['logpdf']:.3f}")
