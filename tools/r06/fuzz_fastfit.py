"""Random dense configurations (tools/fuzz_more.py's generator): the prepared objective against the autograd route, layer by layer, at the
initial point and at a perturbed one: python tools/r06/fuzz_fastfit.py first last"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from tools.fuzz_more import case
from tests.test_fastfit import _layer_objectives
from tests.conftest import make_engine
from gpar_amd.engine import set_engine
from gpar_amd.regression import GPARRegressor
eng = make_engine("hip"); set_engine(eng)
bad = done = skipped = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    kw, x, y, w, xs = case(seed)
    if "x_ind" in kw or x.shape[0] > 4096:
        skipped += 1; continue
    reg = GPARRegressor(**kw)
    reg.condition(x, y, w)
    worst = 0.0
    for pi in range(reg.p):
        try:
            fast, fg, x0 = _layer_objectives(reg, eng, pi, None)
        except AssertionError:
            print(seed, "layer", pi, "prepared objective does not apply"); continue
        rng = np.random.default_rng(seed * 10 + pi)
        for trial in range(2):
            xv = x0 + (0.0 if trial == 0 else 0.25 * rng.standard_normal(x0.shape))
            vf, gf = fast.fg(xv); vr, gr = fg(xv)
            dv = 0.0 if vf == vr else abs(vf - vr) / max(abs(vr), 1.0)
            dg = np.max(np.abs(gf - gr)) / max(np.abs(gr).max(), 1e-3)
            worst = max(worst, dg)
            if not (np.isnan(vf) and np.isnan(vr)) and (dv > 0 or dg > 1e-9):
                bad += 1; print(seed, "layer", pi, "MISMATCH dv %.2e dg %.2e fallbacks %d" % (dv, dg, fast.fallbacks), {k: v for k, v in kw.items() if k != "x_ind"}, x.shape)
        reg.vs.set_vector(x0, fast.names)
    done += 1
    print(seed, x.shape, y.shape[1], "worst gradient difference %.1e" % worst, flush=True)
print("cases", done, "skipped", skipped, "bad", bad)
