"""Dev aid: one case of tools/fuzz_more.py against the oracle engine on several factorisation variants (fused / unfused / first-generation panel /\nwithout the progressive hand-off): a mismatch that all variants share among themselves is conditioning (seed 709: 293 inducing inputs on a line,\ngradient 1e-2 apart whichever arithmetic), one that a single variant shows is a bug.   python tools/dev/fuzz_case_variants.py [seed]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np
from fuzz_more import case
from tests.test_fuzz_parity_gpu import _grads
kw, x, y, w, xs = case(int(sys.argv[1]) if len(sys.argv) > 1 else 709)
ov, og = _grads("oracle", kw, x, y, w)
big = max(np.max(np.abs(og)), 1e-3)
res = {}
for name, env in [("default", {}), ("unfused", {"GPAR_POTRF_FUSED": "0"}), ("panel_v1", {"GPAR_PANEL_V": "1"}), ("noprog", {"GPAR_PANEL_PROGRESSIVE": "0"})]:
    for k, v in env.items(): os.environ[k] = v
    hv, hg = _grads("hip", kw, x, y, w)
    for k in env: del os.environ[k]
    res[name] = (hv, hg)
    print(name, "dv %.2e dg(vs oracle) %.2e" % (abs(hv - ov) / max(abs(ov), 1), np.max(np.abs(hg - og)) / big), flush=True)
names = list(res)
for a in names:
    for b in names:
        if a < b: print(a, b, "dg %.2e" % (np.max(np.abs(res[a][1] - res[b][1])) / big))
