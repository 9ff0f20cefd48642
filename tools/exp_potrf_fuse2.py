"""Two panels per launch (potrf_group2_kernel) against the one-panel-per-launch schedule (development aid): lone and lock-step
factorisations at a few sizes, under GPAR_POTRF_FUSE2_ROWS / GPAR_POTRF_FUSE2_BATCH_ROWS / GPAR_POTRF_LA_SMALL_TILES2.  Prints the time,
the factor's deviation from the unfused-schedule factor, logdet, info, and whether repeated runs return the same bits.

    python tools/exp_potrf_fuse2.py [n[xbatch] ...]      e.g. 16384 4096x4 2048x4 8192x16
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpar_amd import hip

dev = torch.device("cuda:0")
cases = sys.argv[1:] or ["1024", "1600", "2048", "4096", "4096x4", "2048x4", "1024x4", "8192", "16384"]
VARIANTS = [("one panel per launch", {"GPAR_POTRF_FUSE2_ROWS": "0", "GPAR_POTRF_FUSE2_BATCH_ROWS": "0"})]
for spec in os.environ.get("FUSE2_SWEEP", "default").split(","):
    if spec == "default":
        VARIANTS.append(("fuse2 default", {}))
    else:   # rows:batch_rows[:tiles2]
        parts = spec.split(":")
        env = {"GPAR_POTRF_FUSE2_ROWS": parts[0], "GPAR_POTRF_FUSE2_BATCH_ROWS": parts[1]}
        if len(parts) > 2:
            env["GPAR_POTRF_LA_SMALL_TILES2"] = parts[2]
        VARIANTS.append((f"fuse2 {spec}", env))
# FUSE2_ENVS="A=1;B=2,C=3": further variants given as environment assignments
for spec in filter(None, os.environ.get("FUSE2_ENVS", "").split(",")):
    VARIANTS.append((spec, dict(kv.split("=") for kv in spec.split(";"))))
KEYS = sorted({k for _, v in VARIANTS for k in v} | {"GPAR_POTRF_FUSE2_ROWS", "GPAR_POTRF_FUSE2_BATCH_ROWS", "GPAR_POTRF_LA_SMALL_TILES2"})
REPS = int(os.environ.get("FUSE2_REPS", "6"))

for case in cases:
    n, batch = (int(v) for v in case.split("x")) if "x" in case else (int(case), 1)
    g = torch.Generator(device="cpu"); g.manual_seed(n)
    X = torch.rand(n, 4, generator=g, dtype=torch.float64).to(dev)
    N = n + 1
    K0 = hip.alloc_matrix(batch * N, N, dev, zero=True)
    for b in range(batch):
        blk = K0[b * N:(b + 1) * N]
        blk[:n, :n] = torch.exp(-0.5 * torch.cdist(X, X) ** 2 / (0.25 + 0.05 * b)); blk[:n, :n].diagonal().add_(0.1)
        blk[n, :n] = torch.sin((5 + b) * X[:, 0])
    A = hip.alloc_matrix(batch * N, N, dev)
    ref = None
    for name, env in VARIANTS:
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        best, first, stable = 1e9, None, True
        for it in range(REPS):
            A.copy_(K0); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if batch == 1:
                logdet, info = hip.potrf_(A, nf=n)
            else:
                logdet, info = hip.potrf_batch_(A, batch, nf=n)
            e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
            L = torch.tril(A.view(batch, N, A.shape[1])[:, :, :N])
            if first is None: first = L.clone()
            elif not torch.equal(L, first): stable = False
        dev_txt = ""
        if ref is None:
            ref = first
        else:
            d = ((first - ref).abs().amax() / ref.abs().amax()).item()
            dev_txt = f"  vs one-panel: max rel {d:.2e}"
        # residual of the first matrix: L (L^T v) against K v
        v = torch.rand(n, 1, dtype=torch.float64, device=dev)
        L0 = first[0, :n, :n]
        Kv = torch.tril(K0[:n, :n]) @ v + torch.tril(K0[:n, :n], -1).T @ v
        res = ((L0 @ (L0.T @ v) - Kv).abs().max() / Kv.abs().max()).item()
        corner = first[0, n, n].item()
        print(f"n={n:6d} x{batch:<2d} {name:28s} {best:8.3f} ms  {batch * n**3/3/best*1e-9:6.2f} TF  info={info.tolist()} logdet0={logdet[0].item():.10f} "
              f"corner={corner:.10f} res={res:.1e} stable={stable}{dev_txt}", flush=True)
    del K0, A, ref, X, first, L
    torch.cuda.empty_cache()
