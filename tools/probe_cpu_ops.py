"""Dev aid: torch operators of one evaluation that run on CPU tensors (candidates for OpenMP parallel regions, whose
worker threads spin after every region: 128 spinning threads exhaust the container's CPU quota and stall the host)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_flatten
from gpar_amd.engine import HipEngine, set_engine
from tools.run_config import build

eng = HipEngine(seed=1)
set_engine(eng)
cfg, reg, x, y = build(sys.argv[1] if len(sys.argv) > 1 else "C4", eng)
float(reg.logpdf(x, y))
count = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        flat, _ = tree_flatten((args, kwargs or {}))
        ts = [a for a in flat if isinstance(a, torch.Tensor)]
        if ts and all(t.device.type == "cpu" for t in ts):
            count[(str(func), max(t.numel() for t in ts))] += 1
        return func(*args, **(kwargs or {}))


what = sys.argv[2] if len(sys.argv) > 2 else "logpdf"
import numpy as np
xs = np.random.default_rng(5).uniform(0, 1, (512, cfg["m"]))
with Log():
    if what == "logpdf":
        float(reg.logpdf(x, y))
    elif what == "fit":
        reg.fit(x, y, iters=1)
    else:
        reg.condition(x, y)
        reg.predict(xs, num_samples=2)
for (name, numel), c in sorted(count.items(), key=lambda kv: -kv[0][1])[:14]:
    print(f"{c:4d} x {name:45s} max numel {numel}")
print("total CPU-tensor ops:", sum(count.values()))
