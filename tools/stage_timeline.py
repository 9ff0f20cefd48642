"""Per-layer GPU timeline of a pipelined evaluation (events at the start / end of every pipeline stage) plus the host
time at which each stage was enqueued.   python tools/stage_timeline.py C2 [depth]"""
import contextlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from gpar_amd import engine as E
from gpar_amd.engine import HipEngine, set_engine
from tools.run_config import build

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
if len(sys.argv) > 2:
    os.environ["GPAR_LAYER_PIPELINE"] = sys.argv[2]
eng = HipEngine(seed=1)
set_engine(eng)
cfg, reg, x, y = build(name, eng)
for _ in range(3):
    reg.logpdf(x, y)
torch.cuda.synchronize()

records = []
original = E._LayerPipeline.stage


@contextlib.contextmanager
def traced(self, i, *inputs):
    with original(self, i, *inputs):
        s = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        h0 = time.perf_counter()
        e0.record(s)
        yield
        e1.record(s)
        records.append((i, h0, time.perf_counter(), e0, e1))


E._LayerPipeline.stage = traced
for rep in range(3):
    records.clear()
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    start.record()
    v = reg.logpdf(x, y)
    torch.cuda.synchronize()
    wall = 1e3 * (time.perf_counter() - t0)
    print(f"{name} evaluation {rep}: wall {wall:.2f} ms")
    for i, h0, h1, e0, e1 in records:
        print(f"  stage {i}: host enqueue {1e3 * (h0 - t0):6.2f} .. {1e3 * (h1 - t0):6.2f} ms   GPU {start.elapsed_time(e0):6.2f} .. {start.elapsed_time(e1):6.2f} ms")
