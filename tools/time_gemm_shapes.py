"""Rate of the NT update shapes of the stacked triangular solve in predict (rows = S n*, remaining columns, K = block width)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpar_amd import hip
dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 204800
for cols in (8192, 2048):
    for K in (512, 1024, 1536, 2048):
        X = torch.randn(rows, K, dtype=torch.float64, device=dev)
        L = torch.randn(cols, K, dtype=torch.float64, device=dev)
        C = hip.alloc_matrix(rows, cols, dev); C.zero_()
        def run(): hip.gemm(X, L, tb=True, alpha=-1.0, beta=1.0, out=C)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); run(); e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 2
        print(f"rows {rows} cols {cols} K {K}: {ms:.2f} ms  {2.0 * rows * cols * K / ms * 1e-9:.1f} TFLOP/s")
        del X, L, C
