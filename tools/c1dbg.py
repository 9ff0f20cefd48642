import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from tests.test_parity_gpu import CONFIGS, _problem, _on
from gpar_amd.regression import GPARRegressor
kw, n, m, p, missing = CONFIGS["C1-paper-synthetic"]
x, y = _problem(n, m, p, seed=len("C1-paper-synthetic"), missing=missing)
xs = np.random.default_rng(1).uniform(0, 1, (40, m))
def run():
    reg = GPARRegressor(**kw)
    reg.condition(x, y)
    return np.stack(reg.sample(xs, posterior=True, num_samples=3, latent=True))
a = _on("oracle", run); b = _on("hip", run)
print("max abs diff", np.abs(a-b).max(axis=(1,2)), "per output", np.abs(a-b).max(axis=(0,1)))
