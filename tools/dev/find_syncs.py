"""Development aid: where does an evaluation of a BASELINE configuration synchronise with the host?  (torch's sync debug mode)"""
import os, sys, warnings, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
sys.argv = [sys.argv[0]] + sys.argv[1:]
from tools.run_config import CONFIGS, build
from gpar_amd.engine import HipEngine, set_engine
eng = HipEngine(device="cuda:0", seed=1); set_engine(eng)
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
cfg, reg, x, y = build(name, eng)
for _ in range(2):
    float(reg.logpdf(x, y))
torch.cuda.synchronize()
seen = {}
def showwarning(message, category, filename, lineno, file=None, line=None):
    stack = [f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in traceback.extract_stack()[:-2] if "gpar_amd" in f.filename or "run_config" in f.filename]
    key = " <- ".join(reversed(stack[-4:]))
    seen[key] = seen.get(key, 0) + 1
warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
v = reg.logpdf(x, y)
torch.cuda.set_sync_debug_mode("default")
for k, c in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(c, k)
print("value", float(v))
