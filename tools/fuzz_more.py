import os, sys, time, traceback
sys.path.insert(0, os.getcwd())
import numpy as np
from tests.test_fuzz_parity_gpu import _grads, _run
from tests.conftest import make_engine

def case(seed):
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([int(rng.integers(300, 1200)), int(rng.integers(1200, 3000)), int(rng.integers(4608, 5300))], p=[0.5, 0.35, 0.15]))
    m, p = int(rng.integers(1, 3)), int(rng.integers(1, 4))
    kw = dict(scale=float(rng.uniform(0.1, 1.0)), noise=float(10 ** rng.uniform(-3, -0.5)), normalise_y=bool(rng.integers(2)), linear=bool(rng.integers(2)),
              nonlinear=bool(rng.integers(2)), rq=bool(rng.integers(2)), per=bool(rng.integers(4) == 0), input_linear=bool(rng.integers(3) == 0),
              markov=[None, 1, 2][int(rng.integers(3))], impute=bool(rng.integers(2)), replace=bool(rng.integers(3) == 0))
    if rng.integers(2):
        M = int(rng.integers(20, 400))
        kw["x_ind"] = np.linspace(0, 1, M)[:, None] if m == 1 else rng.uniform(0, 1, (M, m))
        kw["sparse_method"] = ["vfe", "vfe", "fitc", "dtc"][int(rng.integers(4))]
    x = rng.uniform(0, 1, (n, m))
    cols = []
    for i in range(p):
        base = np.sin(2 * np.pi * (x @ rng.uniform(0.5, 1.5, m)) + i)
        if cols:
            base = base + 0.4 * cols[-1]
        cols.append(base + 0.1 * rng.standard_normal(n))
    y = np.stack(cols, axis=1) * rng.uniform(0.5, 20) + rng.uniform(-5, 5)
    if rng.integers(2):
        y[rng.random(y.shape) < 0.1] = np.nan
        y[0] = 0.2
    w = None if rng.integers(2) else rng.uniform(0.5, 2.0, (n, p))
    xs = rng.uniform(0, 1, (int(rng.integers(5, 300)), m))
    return kw, x, y, w, xs

if __name__ == "__main__":
    bad = 0
    for seed in range(int(sys.argv[1]), int(sys.argv[2])):
        kw, x, y, w, xs = case(seed)
        desc = {k: (v.shape if hasattr(v, "shape") else v) for k, v in kw.items()}
        t0 = time.time()
        try:
            sparse = "x_ind" in kw
            hv, hg = _grads("hip", kw, x, y, w)
            try:
                ov, og = _grads("oracle", kw, x, y, w)
            except Exception as e:
                print(seed, "ORACLE FAILED but hip ok", type(e).__name__, x.shape, desc); continue
            tol = 1e-5 if sparse else 1e-9
            dv = abs(hv - ov) / max(abs(ov), 1.0)
            big = max(np.max(np.abs(og)), 1e-3)
            dg = np.max(np.abs(hg - og)) / big
            hp, hpost, hs, _, _ = _run("hip", kw, x, y, w, xs)
            op, opost, os_, _, _ = _run("oracle", kw, x, y, w, xs)
            dpost = abs(hpost - opost) / max(abs(opost), 1.0)
            ds = np.max(np.abs(hs - os_)) / max(1.0, np.abs(os_).max())
            flag = "" if (dv <= tol and dg <= (1e-3 if sparse else 1e-6) and dpost <= tol * 10 and ds <= (1e-3 if sparse else 1e-6)) else "  <<<<<< MISMATCH"
            bad += bool(flag)
            print(seed, x.shape, y.shape[1], "sparse" if sparse else "dense", "dv %.1e dg %.1e dpost %.1e dsample %.1e  %.1fs%s" % (dv, dg, dpost, ds, time.time() - t0, flag), desc if flag else "", flush=True)
        except Exception as e:
            bad += 1
            print(seed, "HIP FAILED", type(e).__name__, str(e)[:150], x.shape, desc, flush=True)
    print("bad:", bad)
