"""gpar_trsm_rlt (B <- B L^-T) for tall and short right-hand-side stacks:  python tools/time_trsm_rlt.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpar_amd import hip
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = torch.Generator(device="cpu"); g.manual_seed(n)
X = torch.rand(n, 4, generator=g, dtype=torch.float64).to(dev)
K = hip.alloc_matrix(n, n, dev); K.copy_(torch.exp(-0.5 * torch.cdist(X, X) ** 2 / 0.25)); K.diagonal().add_(0.1)
hip.potrf_(K)
for rows in (2048, 16384, 65536, 204800):
    B = hip.alloc_matrix(rows, n, dev); B.normal_()
    hip.trsm_rlt_(K, B); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); hip.trsm_rlt_(K, B); e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"trsm_rlt rows={rows} n={n}: {ms:.2f} ms  {rows * n * n / ms * 1e-9:.1f} TFLOP/s")
    del B
