import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from gpar_amd.engine import HipEngine, set_engine
eng = HipEngine(seed=1); set_engine(eng)
keep = sys.argv[1].split(",")
for k in list(bench.GRID):
    if k not in keep: del bench.GRID[k]
bench.lone_factorisation_leg = lambda eng: {}
g = bench.config_grid_leg(eng, cpu=(len(sys.argv) > 2))
for k, r in g.items():
    if isinstance(r, dict): print(k, {a: b for a, b in r.items() if a.endswith("_ms") or a.endswith("ms_best") or a.endswith("_all")})
