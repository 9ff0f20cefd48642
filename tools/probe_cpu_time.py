"""Dev aid: CPU time against wall time of fit / predict (spinning OpenMP workers would show as CPU time >> wall time and run
the container into its CPU quota)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synthetic
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor

n, m, p = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 4, 8
eng = HipEngine(seed=1); set_engine(eng)
x, y = synthetic(n, m, p)
xs = np.random.default_rng(5).uniform(0, 1, (1024, m))
reg = GPARRegressor(scale=0.5, linear=True, nonlinear=True, markov=2, noise=0.1, normalise_y=True)


import glob


def ticks():
    out = {}
    for q in glob.glob("/proc/self/task/*/stat"):
        try:
            f = open(q).read()
            rest = f[f.rindex(")") + 2:].split()
            out[q.split("/")[4]] = int(rest[11]) + int(rest[12])
        except Exception:
            pass
    return out


def timed(label, fn):
    torch.cuda.synchronize(); w0, c0, t0 = time.perf_counter(), time.process_time(), ticks()
    fn(); torch.cuda.synchronize()
    t1 = ticks()
    d = sorted((t1[k] - t0.get(k, 0) for k in t1), reverse=True)
    print(f"{label}: wall {time.perf_counter() - w0:.3f} s, cpu {time.process_time() - c0:.3f} s; threads {len(t1)}, busy (10 ms ticks) top {d[:6]}, "
          f"threads with > 0: {sum(1 for v in d if v > 0)}")


timed("fit(iters=3), numpy inputs", lambda: reg.fit(x, y, iters=3))
timed("fit(iters=3) again", lambda: reg.fit(x, y, iters=3))
timed("predict(S=8), numpy inputs", lambda: reg.predict(xs, num_samples=8))
timed("logpdf, numpy inputs", lambda: reg.logpdf(x, y))
timed("logpdf x 5, numpy inputs", lambda: [reg.logpdf(x, y) for _ in range(5)])
print(open("/sys/fs/cgroup/cpu.stat").read().replace("\n", " "))
