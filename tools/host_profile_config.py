"""Host-side (Python / launch) cost of one log-marginal-likelihood evaluation of a BASELINE config: cProfile of the
evaluation with the GPU work still asynchronous, against its wall-clock.  python tools/host_profile_config.py C2"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from gpar_amd.engine import HipEngine, set_engine
from tools.run_config import build


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    eng = HipEngine(seed=1)
    set_engine(eng)
    cfg, reg, x, y = build(name, eng)
    for _ in range(3):
        reg.logpdf(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        reg.logpdf(x, y)
    torch.cuda.synchronize()
    print(f"{name}: {1e3 * (time.perf_counter() - t0) / 5:.2f} ms per evaluation (wall)")
    if os.environ.get("SERIAL"):
        os.environ["GPAR_LAYER_PIPELINE"] = "0"
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(5):
        reg.logpdf(x, y)
    prof.disable()
    torch.cuda.synchronize()
    stats = pstats.Stats(prof)
    stats.sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
