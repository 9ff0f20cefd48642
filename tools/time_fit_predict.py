"""Where do fit and predict spend their time at C3 size? (development aid)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synthetic, c3_regressor
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import _construct_gpar

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
m, p = 4, 8
eng = HipEngine(seed=1); set_engine(eng)
x, y = synthetic(n, m, p)
def tic(): torch.cuda.synchronize(); return time.perf_counter()
reg = c3_regressor()
reg.logpdf(x[:256], y[:256])  # instantiate variables / warm up
# one objective + gradient evaluation of the last layer
xd, yd = eng.tensor(x), eng.tensor(y)
gpar = _construct_gpar(reg, reg.vs, m, p)
design = torch.cat([xd, yd[:, : p - 1]], dim=1)
reg.vs.requires_grad(True)
for rep in range(2):
    for v in reg.vs.get_vars(): v.grad = None
    t0 = tic()
    f, noise = _construct_gpar(reg, reg.vs, m, p).layers[p - 1]()
    from gpar_amd.gp import Obs
    obs = Obs(f(design, noise / torch.ones(n, dtype=torch.float64, device=eng.device)), yd[:, p - 1])
    val = f.measure.logpdf(obs)
    t1 = tic()
    val.backward()
    t2 = tic()
    print(f"layer objective {1e3*(t1-t0):.1f} ms, gradient {1e3*(t2-t1):.1f} ms")
reg.vs.requires_grad(False)
t0 = tic(); reg.condition(x, y); t1 = tic()
xs = np.random.default_rng(5).uniform(0, 1, (1024, m))
s = reg.sample(xs, posterior=True, num_samples=1, latent=True); t2 = tic()
s = reg.sample(xs, posterior=True, num_samples=4, latent=True); t3 = tic()
print(f"condition(store) {1e3*(t1-t0):.1f} ms; sample S=1 (incl. conditioning 8 layers) {1e3*(t2-t1):.1f} ms; S=4 {1e3*(t3-t2):.1f} ms")
