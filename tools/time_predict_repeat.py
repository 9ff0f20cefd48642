"""Is predict's run-to-run spread allocation? fit once, predict three times (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synthetic, c3_regressor
from gpar_amd.engine import HipEngine, set_engine
eng = HipEngine(seed=1); set_engine(eng)
x, y = synthetic(16384, 4, 8)
reg = c3_regressor()
def tic(): torch.cuda.synchronize(); return time.perf_counter()
t0 = tic(); reg.fit(x, y, iters=2); t1 = tic()
print(f"fit {t1-t0:.2f} s; reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB")
xs = np.random.default_rng(5).uniform(0, 1, (1024, 4))
for i in range(3):
    t0 = tic(); m = reg.predict(xs, num_samples=8, latent=True); t1 = tic()
    print(f"predict #{i}: {1e3*(t1-t0):.0f} ms; reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB, allocs {torch.cuda.memory_stats()['num_device_alloc']}")
