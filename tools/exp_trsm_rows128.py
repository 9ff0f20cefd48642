"""128-row tasks in the fused triangular-solve block (trsm_block2_rows128_kernel) against 64-row tasks (development aid):
B <- B L^-T for a few (rows, n), timed under GPAR_TRSM_ROWS128_MIN = 0 (always) / 1 << 30 (never), bits compared.

    python tools/exp_trsm_rows128.py [rows x n ...]      e.g. 65536x1024 16384x4096 100000x512
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpar_amd import hip

dev = torch.device("cuda:0")
cases = sys.argv[1:] or ["65536x1024", "65537x1024", "32768x1024", "16384x1024", "16384x4096", "100000x512", "204800x2048", "4096x4096"]
for case in cases:
    rows, n = (int(v) for v in case.split("x"))
    g = torch.Generator(device="cpu"); g.manual_seed(n)
    X = torch.rand(n, 3, generator=g, dtype=torch.float64).to(dev)
    L = hip.alloc_matrix(n, n, dev)
    L.copy_(torch.exp(-0.5 * torch.cdist(X, X) ** 2 / 0.1)); L.diagonal().add_(0.05)
    _, info = hip.potrf_(L); assert int(info.item()) == 0
    B0 = hip.alloc_matrix(rows, n, dev)
    B0.copy_(torch.randn(rows, n, generator=g, dtype=torch.float64))
    B = hip.alloc_matrix(rows, n, dev)
    out = {}
    for name, v in (("64-row tasks", str(1 << 30)), ("128-row tasks", "0")):
        os.environ["GPAR_TRSM_ROWS128_MIN"] = v
        best = 1e9
        for it in range(5):
            B.copy_(B0); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); hip.trsm_rlt_(L, B); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        out[name] = B.clone()
        print(f"rows={rows:7d} n={n:5d}  {name:14s} {best:8.3f} ms  {rows * n * n / best * 1e-9:6.2f} TFLOP/s", flush=True)
    same = torch.equal(out["64-row tasks"], out["128-row tasks"])
    res = ((out["128-row tasks"][:4096] @ torch.tril(L).T - B0[:4096]).abs().max() / B0[:4096].abs().max()).item()
    print(f"    bits: {'same' if same else 'DIFFER'}   residual of the first 4096 rows {res:.1e}", flush=True)
    os.environ.pop("GPAR_TRSM_ROWS128_MIN", None)
    del L, B0, B, out
    torch.cuda.empty_cache()
