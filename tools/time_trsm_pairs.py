import os, sys
sys.path.insert(0, os.getcwd())
import torch
from gpar_amd import hip
dev = torch.device("cuda:0")
for n, rows in ((1024, 65536), (1024, 4097), (1300, 3000), (2048, 20000)):
    g = torch.Generator().manual_seed(n)
    L = hip.alloc_matrix(n, n, dev)
    L.copy_(torch.tril(torch.rand(n, n, generator=g, dtype=torch.float64) * 0.01).to(dev)); L.diagonal().add_(1.0)
    B0 = torch.randn(rows, n, dtype=torch.float64, device=dev)
    res = {}
    for pairs in ("0", "1"):
        os.environ["GPAR_TRSM_PAIRS"] = pairs
        best = 1e9
        for _ in range(4):
            B = hip.alloc_matrix(rows, n, dev); B.copy_(B0); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); hip.trsm_rlt_(L, B); e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1))
        res[pairs] = (best, B.clone())
    print(f"trsm n={n} rows={rows}: pairs off {res['0'][0]:.3f} ms, on {res['1'][0]:.3f} ms, identical bits: {torch.equal(res['0'][1], res['1'][1])}", flush=True)
