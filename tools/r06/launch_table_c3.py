"""Per-launch table of the trailing updates of ONE lock-step C3 evaluation (8 x 16385, batch = 8), from the library's hook:
    python tools/r06/launch_table_c3.py > profiles/r06_c3_launch_table.txt"""
import ctypes, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import synthetic, c3_regressor
from gpar_amd import _lib
from gpar_amd.engine import HipEngine, set_engine
from gpar_amd.regression import _construct_gpar
eng = HipEngine(seed=1); set_engine(eng)
lib = _lib.load()
n, m, p = 16384, 4, 8
x_np, y_np = synthetic(n, m, p)
reg = c3_regressor()
x, y = eng.tensor(x_np), eng.tensor(y_np)
w = torch.ones_like(y)
gpar = _construct_gpar(reg, reg.vs, m, p)
for _ in range(2):
    float(gpar.logpdf(x, y, w))
dump = tempfile.mktemp()
os.environ["GPAR_PROFILE_DUMP"] = dump
lib.gpar_profile_read(None, None, None, None, 1)
lib.gpar_profile_enable(1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); float(gpar.logpdf(x, y, w)); e1.record(); e1.synchronize()
lib.gpar_profile_enable(0)
l, ms, busy, fl = ctypes.c_int(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
lib.gpar_profile_read(ctypes.byref(l), ctypes.byref(ms), ctypes.byref(busy), ctypes.byref(fl), 1)
print(f"== C3 lock-step evaluation: {e0.elapsed_time(e1):.2f} ms; {l.value} update launches, sum {ms.value:.2f} ms, union {busy.value:.2f} ms, "
      f"{fl.value / (busy.value * 1e-3) * 1e-12:.1f} TF over the union")
print("   start_ms    dur_us     rows   cols  Kxbatch   tiles(x8)  rounds   TFLOP/s(own duration)")
def tiles(rows, cols):
    tm, tn = -(-rows // 128), -(-cols // 128); tn = min(tn, tm)
    return tn * (tn + 1) // 2 + (tm - tn) * tn
for line in open(dump):
    if line.startswith("#"): continue
    t0, t1, rows, cols, kb = line.split(); t0, t1, rows, cols, kb = float(t0), float(t1), int(rows), int(cols), int(kb)
    flops = 2.0 * kb * (cols * (cols + 1) * 0.5 + (rows - cols) * cols)
    T = 8 * tiles(rows, cols)
    print(f"{t0:11.3f} {1e3 * (t1 - t0):9.1f} {rows:8d} {cols:6d} {kb:8d} {T:9d} {T / 512:7.2f} {flops / ((t1 - t0) * 1e-3) * 1e-12:9.1f}")
