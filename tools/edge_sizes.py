import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import GPARRegressor
set_engine(HipEngine(seed=1))
rng=np.random.default_rng(0)
for n in [1,2,3,63,64,65]:
    x=rng.uniform(0,1,(n,1)); y=rng.standard_normal((n,2))
    reg=GPARRegressor(nonlinear=True, linear=True, noise=0.1)
    v=float(reg.logpdf(x,y)); reg.condition(x,y)
    s=reg.sample(x,posterior=True); m=reg.predict(x,num_samples=3)
    print(n, round(v,6), s.shape, m.shape, np.isfinite(s).all())
# all-missing column in one row, single output
x=rng.uniform(0,1,(10,1)); y=rng.standard_normal((10,1)); y[3,0]=np.nan
reg=GPARRegressor(nonlinear=True); print("p=1 missing", float(reg.logpdf(x,y)))
reg.fit(x,y,iters=2); print("fit ok")
