"""Gram-build timing on the layer kernels of the BASELINE configs (the HBM-bound kernel north_star names): the widest
layer's symmetric lower-triangle build for C2 / C3 / C5 and the n x M cross-Gram of C4, alone on the GPU.
Prints one JSON line per kernel: algorithmic bytes (8 per stored entry) / time against the 8 TB/s HBM peak."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from gpar_amd import hip
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.kernels import compile_kernel
from gpar_amd.regression import _construct_gpar
from tools.run_config import CONFIGS, build

HBM_PEAK_TBS = 8.0
eng = HipEngine(seed=1)
set_engine(eng)


def timeit(fn, reps=120):
    """`reps` back-to-back launches with an event between consecutive ones (the queue stays full: a launch costs the host ~10 us,
    the kernel 30-300 us).  The kernel loads the fp64 vector ALUs and the HBM write path together, and the chip's power management
    answers within ~2 ms: the first launches after idle run at ~1.98 GHz, then the clock falls as low as 1.25 GHz and recovers to
    1.6-1.85 GHz over the next ~40 launches (tools/exp_gram/harness.hip prints the clock per launch).  Reported: `burst` = mean
    of launches 2-6 after idle, `sustained` = mean of the last 20, `best`."""
    fn()
    torch.cuda.synchronize()
    import time
    time.sleep(0.2)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]
    return {"burst": float(np.mean(ms[1:6])), "sustained": float(np.mean(ms[-20:])), "best": float(min(ms))}


out = []
for name in sys.argv[1:] or ["C2", "C3", "C4", "C5"]:
    cfg, reg, x, y = build(name, eng)
    n, m, p = cfg["n"], cfg["m"], cfg["p"]
    with torch.no_grad():
        reg.logpdf(x[:64], y[:64])  # instantiate the hyper-parameters
    f, noise = _construct_gpar(reg, reg.vs, m, p).layers[p - 1]()
    design = torch.cat([x, y[:, : p - 1]], dim=1)
    ck = compile_kernel(f.kernel, design.shape[1])
    z = hip.featurize(ck, design)
    if "M" in cfg:
        M = cfg["M"]
        zu = hip.featurize(ck, torch.cat([eng.tensor(reg.x_ind), torch.randn(M, p - 1, dtype=torch.float64, device=eng.device)], dim=1))
        K = hip.alloc_matrix(n, M, eng.device)
        rs = torch.full((n,), 3.0, dtype=torch.float64, device=eng.device)
        ms = timeit(lambda: hip.gram(ck, z, zu, out=K, row_scale=rs))
        nbytes, kind = 8.0 * n * M, f"cross n x M = {n} x {M}"
    else:
        K = hip.alloc_matrix(n + 1, n + 1, eng.device)[:n, :n]   # as logpdf builds it: inside the augmented (n + 1) matrix, rows 16 doubles apart from a power of two
        d = torch.full((n,), 0.1, dtype=torch.float64, device=eng.device)
        ms = timeit(lambda: hip.gram(ck, z, None, out=K, lower=True, diag_add=d, diag_const=1e-12))
        nbytes, kind = 8.0 * n * (n + 1) / 2, f"symmetric lower, n = {n}"
    terms = [[fa.type for fa in t.factors] for t in ck.kernel.terms]
    rec = {"config": name, "kernel": kind, "layer": p - 1, "feature_dims": int(ck.dz), "terms": terms, "ms": ms["sustained"], "ms_burst": ms["burst"],
           "ms_best": ms["best"], "algorithmic_bytes": nbytes}
    for key in ("sustained", "burst"):
        rec["tb_per_s_" + key] = nbytes / ms[key] * 1e-9
        rec["frac_of_hbm_peak_" + key] = nbytes / ms[key] * 1e-9 / HBM_PEAK_TBS
    print(json.dumps(rec), flush=True)
    del K, z
