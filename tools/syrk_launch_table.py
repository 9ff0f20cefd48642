"""Per-launch rate of the trailing updates of ONE gpar_potrf (n = 16384, look-ahead off: every launch alone on the chip):
    rocprofv3 --kernel-trace -f csv -d D -o kt -- python tools/syrk_launch_table.py run ;  python tools/syrk_launch_table.py table D"""
import csv, glob, os, sys
if sys.argv[1] == "run":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from gpar_amd import hip
    n = 16384
    os.environ["GPAR_POTRF_LOOKAHEAD"] = "0"
    g = torch.Generator(device="cpu"); g.manual_seed(1)
    X = torch.rand(n, 4, generator=g, dtype=torch.float64).cuda()
    K = hip.alloc_matrix(n + 1, n + 1, X.device); K.zero_()
    K[:n, :n] = torch.exp(-0.5 * torch.cdist(X, X) ** 2 / 0.25); K[:n, :n].diagonal().add_(0.1)
    for _ in range(2):
        A = hip.alloc_matrix(n + 1, n + 1, X.device); A.copy_(K)
        torch.cuda.synchronize()
        hip.potrf_(A, nf=n); torch.cuda.synchronize()
else:
    rows = []
    for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
    rows.sort()
    # the last factorisation: kernels after the last potrf_zero_flags_kernel
    last = max(i for i, r in enumerate(rows) if "potrf_zero_flags" in r[2])
    sel = rows[last:]
    t0 = sel[0][0]
    tot_syrk = tot_panel = 0.0
    print("  start_ms   dur_us   grid(WGs)  kernel")
    for s, e, name, g in sel:
        short = name.split("(")[0].replace("gpar::", "").replace("void ", "")[-34:]
        d = (e - s) * 1e-3
        if "gemm_f64" in name: tot_syrk += d
        if "panel2" in name: tot_panel += d
        print(f"{1e-6 * (s - t0):9.3f} {d:9.1f} {g // 256:9d}  {short}")
    print(f"span {1e-6 * (sel[-1][1] - t0):.3f} ms; trailing updates {tot_syrk * 1e-3:.3f} ms; panel kernels {tot_panel * 1e-3:.3f} ms")
