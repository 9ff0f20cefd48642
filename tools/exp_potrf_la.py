"""Look-ahead policy of a lone factorisation (development aid): time gpar_potrf at a few sizes under the round-3 switches
GPAR_POTRF_LA_SPLIT / GPAR_POTRF_REST_AFTER_LA / GPAR_POTRF_PAIR_ROWS / GPAR_POTRF_GROUP, and check that the factor keeps its bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpar_amd import hip

dev = torch.device("cuda:0")
sizes = [int(a) for a in sys.argv[1:]] or [16384, 8192]
VARIANTS = [
    ("round-2 schedule", {"GPAR_POTRF_LA_SPLIT": "0", "GPAR_POTRF_REST_AFTER_LA": "0"}),
    ("split", {"GPAR_POTRF_LA_SPLIT": "1", "GPAR_POTRF_REST_AFTER_LA": "0"}),
    ("split + rest-after-la 8192", {"GPAR_POTRF_REST_AFTER_LA": "8192"}),
    ("split + rest-after-la 10240 (default)", {}),
    ("panel split from 6144 rows", {"GPAR_POTRF_REST_AFTER_LA": "0", "GPAR_POTRF_PANEL_SPLIT_ROWS": "6144"}),
    ("panel split from 9216 rows", {"GPAR_POTRF_REST_AFTER_LA": "0", "GPAR_POTRF_PANEL_SPLIT_ROWS": "9216"}),
    ("panel split from 11264 rows", {"GPAR_POTRF_REST_AFTER_LA": "0", "GPAR_POTRF_PANEL_SPLIT_ROWS": "11264"}),
    ("panel split from 13312 rows", {"GPAR_POTRF_REST_AFTER_LA": "0", "GPAR_POTRF_PANEL_SPLIT_ROWS": "13312"}),
    ("panel split always", {"GPAR_POTRF_REST_AFTER_LA": "0", "GPAR_POTRF_PANEL_SPLIT_ROWS": "0"}),
]
KEYS = sorted({k for _, v in VARIANTS for k in v})
for n in sizes:
    g = torch.Generator(device="cpu"); g.manual_seed(n)
    X = torch.rand(n, 4, generator=g, dtype=torch.float64).to(dev)
    K0 = hip.alloc_matrix(n + 1, n + 1, dev, zero=True)
    K0[:n, :n] = torch.exp(-0.5 * torch.cdist(X, X) ** 2 / 0.25); K0[:n, :n].diagonal().add_(0.1)
    K0[n, :n] = torch.sin(5 * X[:, 0])
    A = hip.alloc_matrix(n + 1, n + 1, dev)
    ref = None
    for name, env in VARIANTS:
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        best = 1e9
        for it in range(4):
            A.copy_(K0); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); _, info = hip.potrf_(A, nf=n); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        assert int(info.item()) == 0
        L = torch.tril(A)
        same = "" if ref is None else ("  bits: same" if torch.equal(L, ref) else f"  bits: DIFFER (max rel {((L - ref).abs().max() / ref.abs().max()).item():.2e})")
        if ref is None: ref = L.clone()
        print(f"n={n:6d}  {name:42s} {best:8.3f} ms  {n**3/3/best*1e-9:6.2f} TFLOP/s{same}", flush=True)
        del L
    del K0, A, ref, X
    torch.cuda.empty_cache()
