"""Run the log marginal likelihood of one BASELINE.json configuration a few times (for rocprofv3 / timing).

    python tools/run_config.py C4 [--evals 3] [--warmup 1] [--serial]

Prints one JSON line: wall-clock per evaluation (ms), the value, and the algorithmic flop count of the configuration
 with the fraction of the fp64 matrix peak the wall-clock corresponds to.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from bench import synthetic
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor

PEAK_TFLOPS = 78.6

CONFIGS = {
    "C2": dict(n=4096, m=2, p=4, kw=dict(scale=0.5, linear=True, nonlinear=False, noise=0.1)),
    "C3": dict(n=16384, m=4, p=8, kw=dict(scale=0.5, linear=True, nonlinear=True, markov=2, noise=0.1)),
    "C4": dict(n=65536, m=8, p=4, kw=dict(scale=0.5, linear=True, nonlinear=True, noise=0.1), M=1024),
    "C5": dict(n=8192, m=3, p=16, kw=dict(scale=0.5, per=True, rq=True, linear=True, nonlinear=True, noise=0.1)),
}


def algorithmic_flops(cfg):
    n, p = cfg["n"], cfg["p"]
    if "M" in cfg:
        M = cfg["M"]
        return p * (2.0 * M * M * n + 2.0 * M**3 / 3.0)
    return p * (n**3 / 3.0 + n * n)


def build(name, engine):
    cfg = CONFIGS[name]
    x, y = synthetic(cfg["n"], cfg["m"], cfg["p"])
    kw = dict(cfg["kw"], normalise_y=False)
    if "M" in cfg:
        kw["x_ind"] = np.random.default_rng(3).uniform(0, 1, (cfg["M"], cfg["m"]))
    reg = GPARRegressor(**kw)
    return cfg, reg, engine.tensor(x), engine.tensor(y)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=sorted(CONFIGS))
    ap.add_argument("--evals", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--serial", action="store_true", help="no layer pipelining, no look-ahead (kernels run alone)")
    args = ap.parse_args()
    if args.serial:
        os.environ["GPAR_LAYER_PIPELINE"] = "0"
        os.environ["GPAR_POTRF_LOOKAHEAD"] = "0"
    eng = HipEngine(seed=1)
    set_engine(eng)
    cfg, reg, x, y = build(args.config, eng)
    value = None
    for _ in range(args.warmup):
        value = float(reg.logpdf(x, y))
    times = []
    for _ in range(args.evals):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        value = float(reg.logpdf(x, y))
        torch.cuda.synchronize()
        times.append(1e3 * (time.perf_counter() - t0))
    flops = algorithmic_flops(cfg)
    best = min(times)
    print(json.dumps({
        "config": args.config, "n": cfg["n"], "m": cfg["m"], "p": cfg["p"], "M": cfg.get("M"),
        "logpdf": value, "ms": times, "ms_best": best, "serial": args.serial,
        "algorithmic_flops": flops, "tflops": flops / (best * 1e-3) * 1e-12,
        "frac_of_fp64_matrix_peak": flops / (best * 1e-3) * 1e-12 / PEAK_TFLOPS,
    }))


if __name__ == "__main__":
    main()
