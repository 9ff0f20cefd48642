import os, sys
sys.path.insert(0, os.getcwd())
import torch
from gpar_amd import hip
dev = torch.device("cuda:0")
for n, rows in ((256, 20000), (512, 20000), (768, 20000), (1000, 20000), (512, 2048), (900, 400)):
    g = torch.Generator().manual_seed(n)
    L = hip.alloc_matrix(n, n, dev)
    L.copy_(torch.tril(torch.rand(n, n, generator=g, dtype=torch.float64) * 0.01).to(dev)); L.diagonal().add_(1.0)
    B0 = torch.randn(rows, n, dtype=torch.float64, device=dev)
    res = {}
    for nb in ("64", "512", "1024"):
        os.environ["GPAR_TRSM_NB"] = nb
        best = 1e9
        for _ in range(4):
            B = hip.alloc_matrix(rows, n, dev); B.copy_(B0); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); hip.trsm_rlt_(L, B); e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1))
        res[nb] = (best, B.clone())
    back = {}
    for mn in ("100000", "128"):
        os.environ["GPAR_TRSM_BACK_FUSED_MIN"] = mn
        best = 1e9
        for _ in range(4):
            B = hip.alloc_matrix(rows, n, dev); B.copy_(B0); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); hip.trsm_rln_(L, B); e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1))
        back[mn] = (best, B.clone())
    print(f"   backward: strips {back['100000'][0]:.3f} ms, fused {back['128'][0]:.3f} ms, max diff {float((back['100000'][1] - back['128'][1]).abs().max()):.1e}")
    d = float((res["64"][1] - res["512"][1]).abs().max())
    print(f"trsm n={n} rows={rows}: NB=64 {res['64'][0]:.3f} ms, NB=512 {res['512'][0]:.3f} ms, NB=1024 {res['1024'][0]:.3f} ms, max diff {d:.1e}", flush=True)
