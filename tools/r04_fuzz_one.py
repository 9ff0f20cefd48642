"""Re-run single seeds of tools/fuzz_more.py (value + gradient on the HIP path only), for bisecting with switches."""
import os, sys
sys.path.insert(0, os.getcwd())
seeds = [int(s) for s in sys.argv[1:]]
sys.argv = ["x", "0", "0"]
exec(open("tools/fuzz_more.py").read().split("bad = 0")[0])
from tests.test_fuzz_parity_gpu import _grads
for seed in seeds:
    kw, x, y, w, xs = case(seed)
    try:
        hv, hg = _grads("hip", kw, x, y, w)
        print(seed, "hip ok", hv, flush=True)
    except Exception as e:
        print(seed, "hip FAILED", type(e).__name__, str(e)[:80], flush=True)
