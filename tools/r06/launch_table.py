"""Per-launch rate of the trailing updates of ONE gpar_potrf, from the library's own hook (GPAR_PROFILE_DUMP: hipEvents around every
counted update launch, with its shape): tiles, rounds of 512 workgroup slots, duration, TFLOP/s - with look-ahead off (every launch
alone on the chip) and on (the schedule as it runs).

    python tools/r06/launch_table.py [n] > profiles/r06_potrf_launch_table_n16384.txt"""
import ctypes, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gpar_amd import _lib, hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
lib = _lib.load()
g = torch.Generator(device="cpu"); g.manual_seed(1)
X = torch.rand(n, 4, generator=g, dtype=torch.float64).cuda()
K = hip.alloc_matrix(n + 1, n + 1, X.device); K.zero_()
K[:n, :n] = torch.exp(-0.5 * torch.cdist(X, X) ** 2 / 0.25); K[:n, :n].diagonal().add_(0.1)
K[n, :n] = torch.sin(5 * X[:, 0])
A = hip.alloc_matrix(n + 1, n + 1, X.device)


def tiles(rows, cols):
    tm, tn = -(-rows // 128), -(-cols // 128)
    tn = min(tn, tm)
    return tn * (tn + 1) // 2 + (tm - tn) * tn


for la in ("0", "1"):
    os.environ["GPAR_POTRF_LOOKAHEAD"] = la
    dump = tempfile.mktemp()
    best = None
    for rep in range(4):
        A.copy_(K)
        torch.cuda.synchronize()
        if os.path.exists(dump):
            os.remove(dump)
        os.environ["GPAR_PROFILE_DUMP"] = dump
        lib.gpar_profile_read(None, None, None, None, 1)
        lib.gpar_profile_enable(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.potrf_(A, nf=n)
        e1.record(); e1.synchronize()
        lib.gpar_profile_enable(0)
        l, ms, busy, fl = ctypes.c_int(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        lib.gpar_profile_read(ctypes.byref(l), ctypes.byref(ms), ctypes.byref(busy), ctypes.byref(fl), 1)
        total = e0.elapsed_time(e1)
        if rep > 0 and (best is None or total < best[0]):
            best = (total, open(dump).read(), l.value, ms.value, busy.value, fl.value)
    del os.environ["GPAR_PROFILE_DUMP"]
    total, text, l, ms, busy, fl = best
    print(f"== gpar_potrf n = {n} (augmented {n + 1}), look-ahead {'on' if la == '1' else 'off'}: {total:.3f} ms (events add ~{l * 0.01:.2f} ms); "
          f"{l} update launches, sum of durations {ms:.3f} ms, union {busy:.3f} ms, {fl / (busy * 1e-3) * 1e-12:.1f} TF over the union, "
          f"{fl / (ms * 1e-3) * 1e-12:.1f} TF per launch")
    print("   start_ms    dur_us     rows   cols     K   tiles  rounds   TFLOP/s")
    for line in text.splitlines():
        if line.startswith("#"):
            continue
        t0, t1, rows, cols, kb = line.split()
        t0, t1, rows, cols, kb = float(t0), float(t1), int(rows), int(cols), int(kb)
        flops = 2.0 * kb * (cols * (cols + 1) * 0.5 + (rows - cols) * cols)
        T = tiles(rows, cols)
        print(f"{t0:11.3f} {1e3 * (t1 - t0):9.1f} {rows:8d} {cols:6d} {kb:5d} {T:7d} {T / 512:7.2f} {flops / ((t1 - t0) * 1e-3) * 1e-12:9.1f}")
