"""Headline benchmark: GPAR log-marginal-likelihood throughput at BASELINE.json's C3 configuration
(n = 16384, m = 4, p = 8, nonlinear + linear output dependencies, markov = 2, dense Cholesky, fp64), layers sharded
over the GPUs of one node (one process per GPU; RCCL only for the 8-byte sum of layer log-likelihoods).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 16384 --m 4 --p 8] [--no-extras] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one evaluation of the model's log marginal likelihood on one synthetic data set whose inputs are
already resident in HBM (Gram build -> augmented Cholesky -> quadratic form, for each of the p layers).  Rank 0
prints ONE JSON line.  Besides the contract fields it carries
  roofline      the trailing-update SYRK of the blocked Cholesky (the dominant kernel): algorithmic flops
                (rem * (rem + 1) * kb per launch) / hipEvent-measured time, against the fp64 matrix peak;
  cpu_baseline  the reference's CPU path restated on torch-CPU fp64 operators (oracle/torch_cpu.py, kind "port"), timed
                on this box's host cores on a bounded sample of the same workload;
  fit_predict   wall-clock of `fit(iters=20)` + `predict(num_samples=100)` at the same size (the other half of BASELINE's metric);
  config_grid   log marginal likelihood wall-clock of BASELINE.json's other configurations (C1, C2, C4, C5; C2 / C5 with their
                predict legs), each with the torch-CPU port timed beside it (`cpu_baseline`: C1 and C2 in full, C4 and C5 one
                full-size layer x p), the lone factorisation n = 1024 .. 16384, and `p1_ms_per_step` (the evaluation one rank
                runs at N = 8);
  small_n       the small-problem crossover: logpdf and fit(iters=20) at n = 100, 400, 1024, 2048 (m = 2, p = 4), GPU and CPU.
BASELINE.md section 4's table can be filled from this one line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

PROCESS_T0 = time.perf_counter()   # wall clock of the whole run, printed as `wall_s` (the driver's own clock also sees interpreter start-up)

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MATRIX_PEAK_TFLOPS = 78.6  # MI355X spec (256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz); ubench ceiling 71.3 (profiles/)


def synthetic(n, m, p, seed=1234):
    """Smooth chained outputs in the style of examples/paper/synthetic.py (each output depends nonlinearly on the
    inputs and on the previous outputs), standardised; no missing values."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(0.0, 1.0, (n, m))
    cols = []
    for i in range(p):
        f = np.sin(2 * np.pi * (x @ rng.uniform(0.5, 1.5, m)) + 0.7 * i) / (1.0 + x[:, i % m])
        if i >= 1:
            f = f + np.cos(cols[-1]) ** 2
        if i >= 2:
            f = f + 0.5 * cols[-2] * cols[-1]
        f = f + 0.1 * rng.standard_normal(n)
        cols.append((f - f.mean()) / f.std())
    return x, np.stack(cols, axis=1)


def c3_regressor(replace=False):
    from gpar_amd.regression import GPARRegressor

    return GPARRegressor(scale=0.5, linear=True, nonlinear=True, markov=2, noise=0.1, replace=replace, impute=True,
                         normalise_y=False)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", "--rows", dest="n", type=int, default=16384)  # --rows: torchrun's parser chokes on --n
    ap.add_argument("--m", type=int, default=4)
    ap.add_argument("--p", type=int, default=8)
    ap.add_argument("--no-extras", action="store_true", help="skip the fit + predict leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-isolated", action="store_true",
                    help="skip the extra, untimed layer-after-layer evaluation behind roofline.isolated (the PMC passes use this: every "
                         "dispatch of the update kernel in their trace then belongs to the one lock-step evaluation)")
    ap.add_argument("--extras-timeout", type=float, default=600.0,
                    help="seconds the fit + predict / CPU-baseline / teardown part may take before every rank exits (rank 0 prints the line first)")
    ap.add_argument("--cpu-budget", type=float, default=30.0, help="seconds of CPU work the bounded torch-CPU baseline may spend on full-size layers")
    ap.add_argument("--init-timeout", type=float, default=300.0,
                    help="seconds the process group may take to form (and the first collective to return) before every rank gives up")
    ap.add_argument("--launch-check", action="store_true",
                    help="form the process group (gloo, no GPU needed), all-reduce one word, print a JSON line and exit: exercises the launcher")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver calls it: this process becomes the launcher of N ranks (one per GPU) on a free
        # local port; every rank re-enters main() with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set and rank 0 prints the line
        relaunch(args.gpus)

    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # read by the HIP runtime when it initialises (see gpar_amd.engine._hardware_queues)
    import threading

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch with --nproc-per-node {args.gpus}, or without a "
                         f"launcher: `python bench.py --gpus {args.gpus}` starts the ranks itself)")

    # From here until the process group has formed and answered one collective, a watchdog: a rank that cannot reach its peers
    # (RCCL initialisation failure, a peer that died at start-up) would otherwise hang in the first collective.
    def init_failed():
        if rank == 0:
            print(json.dumps({"metric": "logpdf_per_s", "value": None, "n_gpus": world,
                              "error": f"process group of {world} ranks did not form within {args.init_timeout} s"}), flush=True)
        os._exit(3)

    init_watchdog = threading.Timer(args.init_timeout, init_failed)
    init_watchdog.daemon = True
    if world > 1:
        init_watchdog.start()

    import datetime

    import torch
    import torch.distributed as dist

    if args.launch_check:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=args.init_timeout))
            word = torch.tensor([float(rank + 1)], dtype=torch.float64)
            dist.all_reduce(word)
            dist.barrier()
        else:
            word = torch.tensor([1.0], dtype=torch.float64)
        init_watchdog.cancel()
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "sum_of_ranks_plus_one": float(word[0])}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    # development aid: GPAR_BENCH_ONE_GPU=1 lets several ranks share GPU 0 over gloo, to exercise the multi-rank control flow
    # (collectives, sharded fit / predict) on a one-GPU box; timings of such a run mean nothing
    one_gpu_dev = os.environ.get("GPAR_BENCH_ONE_GPU") == "1"
    if one_gpu_dev:
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs the MI355X: there is no CPU fallback for the product path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        limit = datetime.timedelta(seconds=args.init_timeout)
        if one_gpu_dev:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=limit)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, timeout=limit, device_id=torch.device(f"cuda:{local_rank}"))
        # the first collective forms the communicator: do it here, under the start-up watchdog, not inside the first warm-up step
        hello = torch.ones(1, dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(hello)
        torch.cuda.synchronize()
        if int(hello.item()) != world:
            raise SystemExit(f"bench.py: first all-reduce returned {hello.item()} on {world} ranks")
    init_watchdog.cancel()

    from gpar_amd import _lib
    from gpar_amd.engine import HipEngine, set_engine
    from gpar_amd.parallel import sharded_logpdf
    from gpar_amd.regression import _construct_gpar

    eng = HipEngine(device=f"cuda:{local_rank}", seed=1)
    set_engine(eng)
    lib = _lib.load()

    n, m, p = args.n, args.m, args.p
    x_np, y_np = synthetic(n, m, p)
    reg = c3_regressor()
    x = eng.tensor(x_np)
    y = eng.tensor(y_np)
    w = torch.ones_like(y)
    gpar = _construct_gpar(reg, reg.vs, m, p)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    timing = {}

    def step():
        return sharded_logpdf(gpar, x, y, w, timing=timing)

    value = None
    for _ in range(args.warmup):
        value = step()
    lib.gpar_profile_read(None, None, None, None, 1)
    lib.gpar_profile_enable(1)
    barrier()
    timing.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        value = step()
    barrier()
    elapsed = time.perf_counter() - t0
    lib.gpar_profile_enable(0)
    import ctypes

    launches, ms, busy, flops = ctypes.c_int(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    lib.gpar_profile_read(ctypes.byref(launches), ctypes.byref(ms), ctypes.byref(busy), ctypes.byref(flops), 1)
    busy_ms = [1e3 * timing.get("busy_s", 0.0) / max(args.steps, 1)]
    collective_ms = [1e3 * timing.get("collective_s", 0.0) / max(args.steps, 1)]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=eng.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor(busy_ms + collective_ms, dtype=torch.float64, device=eng.device)
        everyone = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        busy_ms = [float(b[0].item()) for b in everyone]
        collective_ms = [float(b[1].item()) for b in everyone]

    out = {
        "metric": "logpdf_per_s",
        "value": args.steps / elapsed,
        "unit": "logpdf/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"C3 dense GPAR log marginal likelihood: n={n} m={m} p={p} markov=2 linear+nonlinear output kernels, "
                        f"noise=0.1, inputs resident in HBM",
            "parallelism": f"layer-parallel x{world} (layer i on rank i mod {world}; 8-byte all-reduce only)"
                           + ("" if world > 1 else "; no N > 1 run of this code exists in the builder's records - the multi-GPU path is exercised "
                                                   "on CPU ranks (gloo) and on 1-rank RCCL only, its scaling is predicted from one box (DESIGN section 6)"),
            "logpdf": float(value),
            "layers_per_rank": [len(range(r, p, world)) for r in range(world)],
            "layers_of_rank": [list(range(r, p, world)) for r in range(world)],
        },
        # wall-clock each rank spent on its own layers per step, up to the collective (the slowest one bounds the step), and in the
        # 8-byte all-reduce (for a fast rank: the wait for the slowest)
        "per_rank_busy_ms": busy_ms,
        "per_rank_collective_ms": collective_ms,
        # is the oracle pinned against the real reference?  tests/golden/reference_cases.json is written by
        # tests/golden/make_reference_golden.py wherever gpar + stheno are installed (not here: no network)
        "parity_pin": parity_pin(),
    }
    if launches.value > 0 and busy.value > 0:
        # The layers of an evaluation are factored in lock-step (one batched launch per trailing update); the narrow
        # look-ahead update of step k + 1 and the rest of step k's run on two streams, so more than one launch is often in
        # flight and each one's own duration covers work of the other.  The kernel's rate is therefore taken over the UNION
        # of the launch intervals: flops / busy time = per-launch flops / (average launch duration / concurrency).
        achieved = flops.value / (busy.value * 1e-3) * 1e-12
        traffic = pmc_traffic(n, m, p)
        out["roofline"] = {
            "kernel": "gpar::gemm_f64_kernel<false, true, 1, 128> and its half-tile form <false, true, 1, 64> for launches of at most 256 tiles (every trailing-update launch of gpar_potrf_batch: rank-512 / rank-1536 SYRK of the rest of the matrix and the narrow look-ahead slices, batched over the layers of the evaluation, blockIdx.z = layer; v_mfma_f64_16x16x4)",
            "bound": "mfma",
            "achieved": achieved,
            "peak": FP64_MATRIX_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / FP64_MATRIX_PEAK_TFLOPS,
            "traffic": None if traffic is None else traffic.get("traffic_bytes_per_launch"),
            "traffic_per_evaluation": None if traffic is None else traffic.get("traffic_bytes_per_evaluation"),
            "traffic_launches_per_evaluation": None if traffic is None else traffic.get("launches"),
            "traffic_source": PMC_TRAFFIC_FILE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 1 --warmup 0 --no-extras "
                              "--no-cpu --no-isolated`; FETCH doubled per the micro-architecture guide; `traffic` = bytes per launch, the mean "
                              "over ALL dispatches of the kernel in one evaluation; null when gemm_f64.h / potrf.h / panel2.h changed since)",
            "launches": launches.value,
            "launches_per_step": launches.value / max(args.steps, 1),
            "flop_per_launch": flops.value / launches.value,
            "avg_launch_ms": ms.value / launches.value,
            "concurrency": ms.value / busy.value,
            "busy_ms": busy.value,
            "rank": 0,
        }

    # The timed region overlaps the trailing SYRK of panel k with the fused factorisation of panel k+1 (look-ahead on a
    # second stream), so the live number above is the kernel's rate WHILE SHARING the chip.  One extra, untimed evaluation with look-ahead and layer pipelining switched off gives the same kernel's
    # per-launch rate when it has the GPU to itself (concurrency 1: flops per launch / average launch duration).
    if not args.no_isolated:  # every rank takes part (the evaluation contains a collective), whether or not it owns a layer
        os.environ["GPAR_POTRF_LOOKAHEAD"] = "0"
        os.environ["GPAR_LAYER_PIPELINE"] = "0"
        try:
            lib.gpar_profile_read(None, None, None, None, 1)
            lib.gpar_profile_enable(1)
            step()
            barrier()
            lib.gpar_profile_enable(0)
            l2, ms2, b2, fl2 = ctypes.c_int(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
            lib.gpar_profile_read(ctypes.byref(l2), ctypes.byref(ms2), ctypes.byref(b2), ctypes.byref(fl2), 1)
            if "roofline" in out and l2.value > 0 and ms2.value > 0:
                iso = fl2.value / (ms2.value * 1e-3) * 1e-12
                out["roofline"]["isolated"] = {"achieved": iso, "frac": iso / FP64_MATRIX_PEAK_TFLOPS, "launches": l2.value,
                                               "avg_launch_ms": ms2.value / l2.value,
                                               "note": "same kernel, one untimed evaluation with GPAR_POTRF_LOOKAHEAD=0 "
                                                       "GPAR_LAYER_PIPELINE=0 (layer after layer, nothing co-running): flops per launch / average launch duration"}
        finally:
            del os.environ["GPAR_POTRF_LOOKAHEAD"]
            del os.environ["GPAR_LAYER_PIPELINE"]
        if "roofline" in out:
            out["roofline"]["note"] = ("live value, measured inside the timed region with hipEvents on the launch streams: flops of all "
                                       "launches / union of their intervals (`concurrency` launches are in flight on average: the narrow look-ahead "
                                       "update overlaps the rest of the previous one, and both co-run with the next panel's factorisation; "
                                       "a launch is batched over the layers of the lock-step evaluation); "
                                       "`avg_launch_ms` is the plain per-launch average that `rocprofv3 --stats` reports; see `isolated` "
                                       "for the kernel alone")

    # The headline is measured by now.  Everything below (the fit + predict leg with its collectives, the CPU baseline, the
    # process-group teardown) runs under a watchdog: should a peer die or a collective hang, every rank leaves after
    # `--extras-timeout` seconds and rank 0 still prints the one JSON line, with the leg marked as timed out.
    printed = []

    def emit():
        if rank == 0 and not printed:
            printed.append(True)
            out["wall_s"] = round(time.perf_counter() - PROCESS_T0, 1)
            print(json.dumps(out), flush=True)

    def bail():
        if not args.no_extras:
            out.setdefault("fit_predict", {"error": f"timed out after {args.extras_timeout} s", "n_gpus": world})
        emit()
        os._exit(0)

    watchdog = threading.Timer(args.extras_timeout, bail)
    watchdog.daemon = True
    watchdog.start()
    small_gpu = None
    if rank == 0 and world == 1 and not args.no_extras and not args.no_cpu and (n, m, p) == (16384, 4, 8):
        try:
            small_gpu = small_n_gpu_leg(eng)   # (right behind the headline: see small_n_leg)
        except Exception as exc:  # noqa: BLE001 - the headline must survive
            small_gpu = {"error": f"{type(exc).__name__}: {exc}"}
    if not args.no_extras:  # every rank takes part (sharded fit / predict contain collectives)
        try:
            leg = fit_predict_leg(eng, x_np, y_np, n, m, p, world)
        except Exception as exc:  # the headline line must survive a failure of the extra leg
            leg = {"error": f"{type(exc).__name__}: {exc}", "n_gpus": world}
        if rank == 0:
            out["fit_predict"] = leg
    if rank == 0 and world == 1 and not args.no_extras and (n, m, p) == (16384, 4, 8):
        # the other BASELINE.json configurations on the same GPU, in the same run (parity-test cases, not the headline)
        del x, y, w
        try:
            out["config_grid"] = config_grid_leg(eng, cpu=False)
        except Exception as exc:
            out["config_grid"] = {"error": f"{type(exc).__name__}: {exc}"}
        try:
            out["config_grid"]["p1_ms_per_step"] = p1_leg(eng, n, m)
        except Exception as exc:
            out["config_grid"]["p1_ms_per_step"] = {"error": f"{type(exc).__name__}: {exc}"}
        try:
            if isinstance(small_gpu, dict) and "error" in small_gpu:
                out["small_n"] = small_gpu
            else:
                out["small_n"] = small_n_leg(eng, small_gpu) if not args.no_cpu else None
        except Exception as exc:
            out["small_n"] = {"error": f"{type(exc).__name__}: {exc}"}
        if not args.no_cpu and "error" not in out["config_grid"]:
            config_grid_cpu_legs(out["config_grid"])   # (every GPU leg of the line has run by now)
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline_leg(x_np, y_np, m, p, budget_s=args.cpu_budget)
    emit()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    watchdog.cancel()


def p1_leg(eng, n, m, evals=5):
    """One layer of the headline workload alone on the GPU - the evaluation ONE RANK runs when the eight layers of C3 are spread
    over eight GPUs (Gram build + gpar_potrf + value; the widest layer: two output columns in its inputs)."""
    import torch

    from gpar_amd.regression import _construct_gpar

    x_np, y_np = synthetic(n, m + 2, 1, seed=4321)   # (m + 2 input columns stand for [x, y_(i-2), y_(i-1)] under markov = 2)
    reg = c3_regressor()
    gpar = _construct_gpar(reg, reg.vs, m + 2, 1)
    x, y = eng.tensor(x_np), eng.tensor(y_np)
    w = torch.ones_like(y)
    for _ in range(2):
        float(gpar.logpdf(x, y, w))
    times = []
    for _ in range(evals):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        float(gpar.logpdf(x, y, w))
        torch.cuda.synchronize()
        times.append(1e3 * (time.perf_counter() - t0))
    flops = float(n) ** 3 / 3.0 + float(n) ** 2
    return {"ms_best": min(times), "ms_median": float(np.median(times)), "frac_of_fp64_matrix_peak": flops / (min(times) * 1e-3) * 1e-12 / FP64_MATRIX_PEAK_TFLOPS,
            "note": "a single dense layer at n = %d with %d input columns: what each rank evaluates per step at N = 8" % (n, m + 2)}


def parity_pin():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "reference_cases.json")
    if not os.path.exists(path):
        return "absent"
    import hashlib

    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def relaunch(gpus):
    """Replace this process by `python -m torch.distributed.run` with `gpus` ranks of this very command on 127.0.0.1 and a free
    port (the form the round-end driver uses for N > 1; here for the plain `python bench.py --gpus N` call)."""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")             # (torchrun would set it, with a warning)
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


PMC_TRAFFIC_FILE = os.path.join("profiles", "r06_bench_pmc_traffic.json")
SCHEDULE_SOURCES = ("gemm_f64.h", "potrf.h", "panel2.h")   # the update kernel and the launch schedule of the factorisation


def _schedule_source_sha():
    import hashlib

    h = hashlib.sha256()
    for name in SCHEDULE_SOURCES:
        with open(os.path.join(ROOT, "gpar_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(n, m, p):
    """HBM bytes of the trailing-SYRK launches of ONE evaluation from the committed PMC passes of this very workload
    (tools/pmc_traffic.py; null for any other workload: counters cannot be collected from inside the timed process) - and null if
    the kernel OR the launch schedule has changed since those passes were taken (the file is stamped with a hash over
    csrc/gemm_f64.h, potrf.h and panel2.h), so that a stale figure is never reported beside new code.  Returns the record."""
    path = os.path.join(ROOT, PMC_TRAFFIC_FILE)
    if (n, m, p) != (16384, 4, 8) or not os.path.exists(path):
        return None
    with open(path) as f:
        rec = json.load(f)
    if rec.get("schedule_source_sha16") != _schedule_source_sha():
        return None
    return rec


def fit_predict_leg(eng, x_np, y_np, n, m, p, world, fit_iters=20, num_samples=100, n_star=2048):
    """`fit` with the fixed L-BFGS-B iteration count SURVEY.md 8(d) prescribes (iters=20) + `predict` (the reference's
    default of 100 joint posterior samples, at n* = 2048 held-out inputs) on the same data - the other half of
    BASELINE.json's metric, at the stated size.  One rank: GPARRegressor.fit / predict.  Several ranks: layer pi is trained
    on rank pi mod N (parallel.sharded_fit, hyper-parameters broadcast afterwards), the conditioning is layer-parallel and
    the posterior samples are split over the ranks and reduced on the device (parallel.sharded_predict)."""
    import torch

    from gpar_amd import optimise
    from gpar_amd.parallel import sharded_fit, sharded_predict

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
            torch.cuda.synchronize()

    reg = c3_regressor()
    xs = np.random.default_rng(5).uniform(0, 1, (n_star, m))
    evals_before = optimise.evaluation_count()
    sync()
    t0 = time.perf_counter()
    fit_mode = "single process"
    if world == 1:
        reg.fit(x_np, y_np, iters=fit_iters)
    else:
        fit_mode = sharded_fit(reg, x_np, y_np, iters=fit_iters)
    sync()
    t1 = time.perf_counter()
    if world == 1:
        mean = reg.predict(xs, num_samples=num_samples, latent=True)
    else:
        mean = sharded_predict(reg, xs, num_samples=num_samples, latent=True)   # (samples stay on the device; gpar_sample_stats)
    sync()
    t2 = time.perf_counter()
    evaluations = optimise.evaluation_count() - evals_before
    # algorithmic flops: one training evaluation of one layer = n^3 / 3 (factorisation) + 2 n^3 / 3 (inverse from the
    # factor) = n^3; predict = p conditionings (n^3 / 3 each) + per (layer, sample) n^2 n* + n n*^2 + n*^3 / 3
    fit_flops = evaluations * float(n) ** 3
    predict_flops = p * float(n) ** 3 / 3.0 + p * num_samples * (float(n) ** 2 * n_star + float(n) * n_star**2 + n_star**3 / 3.0)
    leg = {"fit_ms": 1e3 * (t1 - t0), "fit_iters": fit_iters, "fit_evaluations": evaluations,
           "fit_algorithmic_flops": fit_flops,
           "fit_frac_of_fp64_matrix_peak": fit_flops / max(t1 - t0, 1e-9) * 1e-12 / FP64_MATRIX_PEAK_TFLOPS / world,
           "predict_algorithmic_flops": predict_flops,
           "predict_frac_of_fp64_matrix_peak": predict_flops / max(t2 - t1, 1e-9) * 1e-12 / FP64_MATRIX_PEAK_TFLOPS / world,
           "flops_note": "SURVEY 8(d): training evaluation of a layer = n^3 (n^3/3 factorisation + 2n^3/3 inverse), one evaluation per "
                         "L-BFGS-B function call of each layer's own optimisation; predict = p n^3/3 + p S (n^2 n* + n n*^2 + n*^3/3): the "
                         "REFERENCE's algorithm (a triangular solve of n* columns per layer and sample) - fractions are algorithmic "
                         "flops / wall-clock / (78.6 TF x GPUs), whatever the implementation shares between samples: layer 0's inputs are "
                         "the same for every sample, so ONE solve serves all of them and the work actually done is ~(p - 1) / p of the "
                         "predict count (0.97 here means ~0.85 of the matrix peak on the solves that run)",
           "predict_ms": 1e3 * (t2 - t1), "num_samples": num_samples, "n_star": n_star,
           "fit_predict_ms": 1e3 * (t2 - t0), "predict_mean_abs": float(np.mean(np.abs(mean))), "n_gpus": world, "fit_parallelism": fit_mode,
           "timing": "barrier-bracketed wall-clock on rank 0; predict = joint ancestral sampling exactly as the reference's "
                     "predict (conditioning + num_samples posterior draws + Monte-Carlo reduction)"}
    if world == 1 and hasattr(reg, "predict") and "marginal" in reg.predict.__code__.co_varnames:
        sync()
        t3 = time.perf_counter()
        reg.predict(xs, num_samples=num_samples, latent=True, marginal=True)
        sync()
        leg["predict_marginal_ms"] = 1e3 * (time.perf_counter() - t3)
        # SURVEY 8(d)'s second predict series: replace=True (posterior means are fed forward instead of samples: every sample sees
        # the same inputs, one triangular solve of n* rows per layer) - same trained hyper-parameters, conditioning included
        reg_r = c3_regressor(replace=True)
        reg_r.vs = reg.vs.copy(detach=True)
        sync()
        t4 = time.perf_counter()
        reg_r.condition(x_np, y_np)
        reg_r.predict(xs, num_samples=num_samples, latent=True)
        sync()
        leg["predict_replace_ms"] = 1e3 * (time.perf_counter() - t4)
    return leg


GRID = {
    "C1": dict(n=25, m=1, p=3, kw=dict(scale=0.1, linear=True, linear_scale=10.0, nonlinear=True, nonlinear_scale=0.1, noise=0.1)),
    "C2": dict(n=4096, m=2, p=4, kw=dict(scale=0.5, linear=True, nonlinear=False, noise=0.1)),
    "C4": dict(n=65536, m=8, p=4, M=1024, kw=dict(scale=0.5, linear=True, nonlinear=True, noise=0.1)),
    "C5": dict(n=8192, m=3, p=16, kw=dict(scale=0.5, per=True, rq=True, linear=True, nonlinear=True, noise=0.1)),
}


def config_grid_leg(eng, evals=5, warmup=2, cpu=True):
    """Log marginal likelihood of BASELINE.json's other configurations (C2 dense n = 4096, C4 inducing points n = 65536 / M = 1024,
    C5 periodic + RQ n = 8192 p = 16) on one GPU: best and median wall-clock of `evals` evaluations each, with the algorithmic
    flop count against the fp64 matrix peak; C5 also times its second leg, predict(num_samples=200)."""
    import torch

    from gpar_amd.regression import GPARRegressor

    grid = {}
    for name, cfg in GRID.items():
        n, m, p = cfg["n"], cfg["m"], cfg["p"]
        x_np, y_np = paper_synthetic() if name == "C1" else synthetic(n, m, p)
        kw = dict(cfg["kw"], normalise_y=False)
        if "M" in cfg:
            kw["x_ind"] = np.random.default_rng(3).uniform(0, 1, (cfg["M"], m))
        reg = GPARRegressor(**kw)
        x, y = eng.tensor(x_np), eng.tensor(y_np)
        value = None
        for _ in range(warmup):
            value = float(reg.logpdf(x, y))
        times = []
        for _ in range(evals):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            value = float(reg.logpdf(x, y))
            torch.cuda.synchronize()
            times.append(1e3 * (time.perf_counter() - t0))
        flops = p * (2.0 * cfg["M"] ** 2 * n + 2.0 * cfg["M"] ** 3 / 3.0) if "M" in cfg else p * (n**3 / 3.0 + n * n)
        rec = {"n": n, "m": m, "p": p, "M": cfg.get("M"), "logpdf_ms_best": min(times), "logpdf_ms_median": float(np.median(times)),
               "logpdf": value, "algorithmic_flops": flops,
               "frac_of_fp64_matrix_peak": flops / (min(times) * 1e-3) * 1e-12 / FP64_MATRIX_PEAK_TFLOPS}
        if name == "C4":
            # the same bound with the product-first order of A - I offered (GPAR_VFE_SPREAD_MAX: decided on the device from the
            # pivot spread of chol(K_zz); opt-in, it costs digits - HipEngine.vfe_spread_limit)
            os.environ["GPAR_VFE_SPREAD_MAX"] = "1e3"
            try:
                alt = float(reg.logpdf(x, y))
                alt_times = []
                for _ in range(evals):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    alt = float(reg.logpdf(x, y))
                    torch.cuda.synchronize()
                    alt_times.append(1e3 * (time.perf_counter() - t0))
                rec["product_first"] = {"logpdf_ms_best": min(alt_times), "logpdf": alt, "relative_difference": abs(alt - value) / abs(value),
                                        "switch": "GPAR_VFE_SPREAD_MAX=1e3 (off by default)"}
            finally:
                del os.environ["GPAR_VFE_SPREAD_MAX"]
        if name == "C1":
            # BASELINE.json configs[0] AS THE REFERENCE RUNS IT (examples/paper/synthetic.py:37-40): fit with its default iteration
            # limit, then predict(x, num_samples=200, credible_bounds=True, latent=True) on the 200-point grid: fit + predict wall-clock
            from gpar_amd import optimise

            grid_x = np.linspace(0, 1, 200)[:, None]
            runs = []
            for rep in range(3):   # (best of the second and third: the first pays first-use costs)
                trainee = GPARRegressor(**kw)
                before = optimise.evaluation_count()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                trainee.fit(x_np, y_np)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                mean_c1, lo_c1, hi_c1 = trainee.predict(grid_x, num_samples=200, credible_bounds=True, latent=True)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                runs.append((1e3 * (t2 - t0), 1e3 * (t1 - t0), 1e3 * (t2 - t1), optimise.evaluation_count() - before))
            best_run = min(runs[1:])
            rec["fit_predict_ms"], rec["fit_ms"], rec["predict_ms"], rec["fit_evaluations"] = best_run
            rec["fit_predict_ms_all"] = [round(r[0], 1) for r in runs]
            rec["fit_predict_note"] = ("examples/paper/synthetic.py:37-40: fit(x_obs, y_obs) with the default iteration limit (L-BFGS-B to "
                                       "convergence), then predict(x, num_samples=200, credible_bounds=True, latent=True) at 200 inputs")
            rec["predict_finite"] = bool(np.isfinite(mean_c1).all() and np.isfinite(lo_c1).all() and np.isfinite(hi_c1).all())
            del trainee
        if name == "C4":
            # BASELINE.md section 4's other two legs of C4: conditioning (the posterior through PseudoObs, reference model.py:286-287,
            # x_ind gaining a column per layer :298-305) and predict (100 joint samples at 2048 held-out inputs, conditioning included
            # as in the reference's predict)
            from gpar_amd.regression import _construct_gpar

            xs = np.random.default_rng(2).uniform(0, 1, (2048, m))
            cond, preds = [], []
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                reg.condition(x_np, y_np)
                posterior = _construct_gpar(reg, reg.vs, m, p) | (reg.x, reg.y, reg.w)
                torch.cuda.synchronize()
                cond.append(1e3 * (time.perf_counter() - t0))
                del posterior
                t0 = time.perf_counter()
                mean_c4 = reg.predict(xs, num_samples=100 if rep else 4)
                torch.cuda.synchronize()
                preds.append(1e3 * (time.perf_counter() - t0))
            rec["condition_ms"] = min(cond[1:])
            rec["predict_ms"] = min(preds[1:])
            rec["predict_num_samples"], rec["predict_n_star"] = 100, 2048
            rec["predict_finite"] = bool(np.isfinite(mean_c4).all())
            rec["condition_predict_note"] = ("condition = GPARRegressor.condition (host copies, 65536 x 8 inputs from host memory) + the posterior of "
                                             "every layer through PseudoObs (what predict / sample build first); predict = the whole call, "
                                             "conditioning included, 100 joint samples at n* = 2048")
        if name == "C2":
            # the reference's default output dependence (linear only): predict, 100 joint samples at 2048 held-out inputs
            reg.condition(x_np, y_np)
            xs = np.random.default_rng(2).uniform(0, 1, (2048, m))
            reg.predict(xs, num_samples=4)   # (first-use costs of the routine's small operators stay outside the clock)
            import ctypes as _ct

            from gpar_amd import _lib as _l

            def jit_state():
                a, b, c, e, f = _ct.c_int(), _ct.c_int(), _ct.c_int(), _ct.c_int(), _ct.c_int()
                _l.load().gpar_jit_stats(_ct.byref(a), _ct.byref(b), _ct.byref(c))
                _l.load().gpar_aot_stats(_ct.byref(e), _ct.byref(f))
                return {"compiled": a.value, "failed": b.value, "cached": c.value, "archive_entries": e.value, "archive_loaded": f.value}

            runs, jit_before = [], jit_state()
            for _ in range(3):   # (the first run at this sample count may still meet first-use costs - a structure compiled at run
                torch.cuda.synchronize()   # time, the allocator's first blocks of this size: every run is reported, the best one counts)
                t0 = time.perf_counter()
                mean = reg.predict(xs, num_samples=100)
                torch.cuda.synchronize()
                runs.append(1e3 * (time.perf_counter() - t0))
            rec["predict_100_samples_ms"] = min(runs)
            rec["predict_100_samples_ms_all"] = [round(t, 1) for t in runs]
            rec["predict_jit_state"] = {"before": jit_before, "after": jit_state()}
            rec["predict_n_star"] = 2048
            rec["predict_finite"] = bool(np.isfinite(mean).all())
            # training at this size: layer by layer, L-BFGS-B, analytic gradient (second fit of the process: the first one pays the
            # allocator's first big blocks)
            from gpar_amd import optimise

            fits = []
            for rep in range(4):   # (best of three after a first fit that pays first-use costs; four host threads: +-8 % run to run)
                trainee = GPARRegressor(**kw)
                before = optimise.evaluation_count()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                trainee.fit(x_np, y_np, iters=20)
                torch.cuda.synchronize()
                fits.append(1e3 * (time.perf_counter() - t0))
                rec["fit_evaluations"] = optimise.evaluation_count() - before
            rec["fit_20_iters_ms"] = min(fits[1:])
            rec["fit_20_iters_ms_all"] = [round(t, 1) for t in fits]
            rec["fit_finite"] = bool(all(np.all(np.isfinite(v)) for v in trainee.get_variables().values()))
            del trainee
        if name == "C5":
            reg.condition(x_np, y_np)
            xs = np.random.default_rng(2).uniform(0, 1, (2048, m))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mean, lo, hi = reg.predict(xs, num_samples=200, credible_bounds=True)
            torch.cuda.synchronize()
            rec["predict_200_samples_ms"] = 1e3 * (time.perf_counter() - t0)
            rec["predict_n_star"] = 2048
            rec["predict_finite"] = bool(np.isfinite(mean).all() and np.isfinite(lo).all() and np.isfinite(hi).all())
        grid[name] = rec
        del x, y, reg
        torch.cuda.empty_cache()
    grid["lone_factorisation_ms"] = lone_factorisation_leg(eng)
    if cpu:
        config_grid_cpu_legs(grid)
    return grid


def config_grid_cpu_legs(grid):
    """The torch-CPU baseline of every configuration of `grid` (after the GPU legs: see below)."""
    if True:
        # the CPU baselines AFTER every GPU leg: the torch-CPU operators leave an OpenMP team spinning for a while after each parallel
        # region, and in a container with a CPU quota that team competes with the product's own host threads (fit drives up to four)
        for name, cfg in GRID.items():
            n, m, p = cfg["n"], cfg["m"], cfg["p"]
            x_np, y_np = paper_synthetic() if name == "C1" else synthetic(n, m, p)
            kw = dict(cfg["kw"], normalise_y=False)
            if "M" in cfg:
                kw["x_ind"] = np.random.default_rng(3).uniform(0, 1, (cfg["M"], m))
            try:
                grid[name]["cpu_baseline"] = cpu_config_leg(name, cfg, x_np, y_np, kw)
            except Exception as exc:  # noqa: BLE001 - the GPU numbers must survive a failure of the baseline
                grid[name]["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}


def paper_synthetic(seed=1, n=200, noise=0.1):
    """BASELINE.json configs[0]: the data of the reference's examples/paper/synthetic.py:11-23 (three chained outputs, every 8th
    point of a 200-point grid observed: n = 25)."""
    x = np.linspace(0, 1, n)
    f1 = -np.sin(10 * np.pi * (x + 1)) / (2 * x + 1) - x**4
    f2 = np.cos(f1) ** 2 + np.sin(3 * x)
    f3 = f2 * f1**2 + 3 * x
    y = np.stack([f1, f2, f3], axis=1) + noise * np.random.default_rng(seed).standard_normal((n, 3))
    return x[::8, None], y[::8]


def layer_specs(kw, m, p):
    """[(kernel dict, noise)] of the layers of GPARRegressor(**kw) at its initial hyper-parameters, for the torch-CPU port: read off
    the product's host-side model objects (on the numpy engine, no GPU involved)."""
    from gpar_amd.engine import set_engine
    from gpar_amd.regression import GPARRegressor, _construct_gpar
    from oracle import kernels as ok
    from oracle.engine import OracleEngine

    previous = set_engine(OracleEngine())
    try:
        reg = GPARRegressor(**{k: v for k, v in kw.items() if k != "x_ind"})
        gpar = _construct_gpar(reg, reg.vs, m, p)
        out = []
        for pi in range(p):
            f, noise = gpar.layers[pi]()
            out.append((ok.spec_to_dict(f.kernel.resolve(m + pi)), float(noise)))
        return out
    finally:
        set_engine(previous)


def cpu_config_leg(name, cfg, x_np, y_np, kw):
    """The torch-CPU port (oracle/torch_cpu.py, the operators the reference's lab.torch backend dispatches to) on one BASELINE
    configuration: C1 and C2 every layer; C4 (inducing points) and C5 the last, widest layer at full size x p."""
    import torch

    from oracle import torch_cpu as tc

    # (C1's matrices are 25 x 25: every torch thread beyond the first only adds OpenMP hand-offs - in a container with a CPU quota the
    # 16-thread figure is ~30 x the one-thread figure - so the plumbing configuration is timed on ONE core, and says so)
    threads = tc.set_threads(1) if name == "C1" else tc.set_threads()
    n, m, p = cfg["n"], cfg["m"], cfg["p"]
    specs = layer_specs(kw, m, p)
    x_all = torch.as_tensor(np.concatenate([x_np, y_np], axis=1))
    layers = range(p) if name in ("C1", "C2") else [p - 1]
    total, stages_sum = 0.0, {}
    for pi in layers:
        spec, noise = specs[pi]
        design = x_all[:, : m + pi]
        best = None
        for _ in range(3 if n <= 1024 else 1):
            if "M" in cfg:
                value, stages = tc.layer_vfe_bound(spec, design, y_np[:, pi], np.full(n, noise),
                                                   np.concatenate([kw["x_ind"], np.zeros((cfg["M"], pi))], axis=1))
            else:
                value, stages, _ = tc.layer_logpdf(spec, design, y_np[:, pi], np.full(n, noise))
            if best is None or sum(stages.values()) < sum(best.values()):
                best = stages
        total += sum(best.values())
        for k, v in best.items():
            stages_sum[k] = stages_sum.get(k, 0.0) + v
    scale = 1.0 if name in ("C1", "C2") else float(p)
    extra = {}
    if name == "C1":
        # fit + predict as examples/paper/synthetic.py:37-40 on the CPU port: L-BFGS-B to convergence per layer (autograd gradients),
        # then p conditionings + 200 x p posterior draws at the 200-point grid (one draw per layer timed, x 200)
        grid = np.linspace(0, 1, 200)[:, None]
        fit_s, evals, predict_s = 0.0, 0, 0.0
        for pi in range(p):
            spec, noise = specs[pi]
            design = x_all[:, : m + pi]
            _, ev, seconds = tc.layer_fit(spec, noise, design, y_np[:, pi], iters=1000)
            fit_s, evals = fit_s + seconds, evals + ev
            _, stages, L = tc.layer_logpdf(spec, design, y_np[:, pi], np.full(n, noise))
            zz = torch.linalg.solve_triangular(L, torch.as_tensor(y_np[:, pi]).reshape(-1, 1), upper=False)
            star = torch.as_tensor(np.concatenate([grid, np.zeros((200, pi))], axis=1))
            draws = [sum(tc.layer_posterior_sample(spec, design, L, zz, star, np.full(200, noise))[1].values()) for _ in range(5)]
            predict_s += sum(stages.values()) + 200 * min(draws)
        extra = {"fit_predict_ms": 1e3 * (fit_s + predict_s), "fit_ms": 1e3 * fit_s, "predict_ms": 1e3 * predict_s, "fit_evaluations": evals,
                 "fit_predict_sample": "every layer trained to convergence (iters=1000 at most); predict = p conditionings + 200 x (one timed draw per layer)"}
    if name == "C4":
        # conditioning holds the same factors as the bound (L_z, B, A, L_A: the stages above); one posterior draw of the last layer at
        # n* = 2048 timed -> predict = conditioning + p x 100 draws
        spec, noise = specs[p - 1]
        Lz, La, v = tc.layer_vfe_bound.last_factors
        zin = np.concatenate([kw["x_ind"], np.zeros((cfg["M"], p - 1))], axis=1)
        star = torch.as_tensor(np.random.default_rng(2).uniform(0, 1, (2048, m + p - 1)))
        _, st = tc.layer_vfe_posterior_sample(spec, zin, Lz, La, v, star, np.full(2048, noise))
        draw_s = sum(st.values())
        extra = {"condition_ms": 1e3 * total * scale, "predict_ms": 1e3 * (total * scale + p * 100 * draw_s), "per_layer_per_sample_ms": 1e3 * draw_s,
                 "condition_predict_sample": "condition = the bound's factors (one layer x p); predict = that + p x 100 x one timed posterior "
                                             "draw of the last layer at n* = 2048 (cross-Gram, two solves against the M x M factors, n* x n* Cholesky)"}
    if name in ("C2", "C5"):
        # per-unit costs of the other legs on the CPU port, the last (widest) layer: one posterior draw at n* = 2048 (predict = p
        # conditionings + p x S draws) and, for C2, one objective + autograd-gradient evaluation (fit(iters=20) = that x the number of
        # evaluations the GPU's L-BFGS-B made: `fit_evaluations`)
        spec, noise = specs[p - 1]
        design = x_all[:, : m + p - 1]
        _, _, L = tc.layer_logpdf(spec, design, y_np[:, p - 1], np.full(n, noise))
        zz = torch.linalg.solve_triangular(L, torch.as_tensor(y_np[:, p - 1]).reshape(-1, 1), upper=False)
        star = torch.as_tensor(np.random.default_rng(2).uniform(0, 1, (2048, m + p - 1)))
        _, st = tc.layer_posterior_sample(spec, design, L, zz, star, np.full(2048, noise))
        draw_s = sum(st.values())
        S = 100 if name == "C2" else 200
        extra = {"predict_ms": 1e3 * (total * scale + p * S * draw_s), "per_layer_per_sample_ms": 1e3 * draw_s, "predict_num_samples": S,
                 "predict_sample": f"p conditionings (the logpdf figure) + p x {S} x one timed posterior draw of the last layer at n* = 2048"}
        if name == "C2":
            del L, zz
            leaves, rebuild = tc.leaf_spec(spec)
            leaves.append(torch.tensor(noise, dtype=torch.float64))
            _, _, stg = tc.layer_objective_and_gradient(lambda ps: rebuild(ps[:-1]), leaves, design, y_np[:, p - 1], noise_index=-1)
            extra["fit_evaluation_ms"] = 1e3 * (stg["forward_s"] + stg["backward_s"])
            extra["fit_sample"] = "one objective + autograd gradient of the last layer; fit(iters=20) = this x the GPU run's evaluation count"
    return {"logpdf_ms": 1e3 * total * scale, "cores": threads, "kind": "port", **extra,
            "sample": ("every layer, full size" if scale == 1.0 else f"the last (widest) layer at full size x p = {p}"
                       + ("; inducing inputs of that layer: the given ones extended by zero columns" if "M" in cfg else "")),
            "stages_ms": {k: 1e3 * v * scale for k, v in stages_sum.items()}}


SMALL_N_KW = dict(scale=0.5, linear=True, nonlinear=False, noise=0.1, normalise_y=False)


def small_n_gpu_leg(eng, sizes=(100, 400, 1024, 2048), m=2, p=4, iters=20):
    """The GPU half of `small_n`: {n: (logpdf ms, best fit ms, every fit's ms)}."""
    import torch

    from gpar_amd.regression import GPARRegressor

    kw = SMALL_N_KW
    gpu = {}
    for n in sizes:
        x_np, y_np = synthetic(n, m, p)
        x, y = eng.tensor(x_np), eng.tensor(y_np)
        reg = GPARRegressor(**kw)
        for _ in range(3):
            reg.logpdf(x, y)
        times = []
        for _ in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            float(reg.logpdf(x, y))
            torch.cuda.synchronize()
            times.append(1e3 * (time.perf_counter() - t0))
        fits = []
        for _ in range(4):   # (best of the second to fourth fit of a size: the first pays first-use costs)
            trainee = GPARRegressor(**kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            trainee.fit(x_np, y_np, iters=iters)
            torch.cuda.synchronize()
            fits.append(1e3 * (time.perf_counter() - t0))
        gpu[n] = (min(times), min(fits[1:]), [round(t, 1) for t in fits])
        del x, y
    return gpu


def small_n_leg(eng, gpu=None, sizes=(100, 400, 1024, 2048), m=2, p=4, iters=20):
    """The small-problem regime (where GPAR is used most: tens to a few thousand observations): logpdf and fit(iters=20) on the
    GPU and with the torch-CPU port, same data and initial hyper-parameters (C2's model: linear output dependence).  `gpu`: the GPU
    half, measured earlier in the run (small_n_gpu_leg; main() takes it right behind the headline: a problem this small is a chain of
    short kernels, whose duration follows the clock the chip is running at - after the minute of sustained matrix-core load of the
    other legs the same fits take 10-20 % longer); the CPU legs run after every GPU leg of the line (their OpenMP teams keep spinning
    after a parallel region and would compete with fit's host threads)."""
    import torch

    from oracle import torch_cpu as tc

    kw = SMALL_N_KW
    if gpu is None:
        gpu = small_n_gpu_leg(eng, sizes, m, p, iters)
    rows = []
    threads = tc.set_threads()
    for n in sizes:
        x_np, y_np = synthetic(n, m, p)
        specs = layer_specs(kw, m, p)
        x_all = torch.as_tensor(np.concatenate([x_np, y_np], axis=1))
        cpu_logpdf = 0.0
        for pi in range(p):
            best = min(sum(tc.layer_logpdf(specs[pi][0], x_all[:, : m + pi], y_np[:, pi], np.full(n, specs[pi][1]))[1].values())
                       for _ in range(3))
            cpu_logpdf += best
        fit_layers = range(p) if n <= 400 else [p - 1]
        cpu_fit, cpu_evals = 0.0, 0
        for pi in fit_layers:
            _, evals, seconds = tc.layer_fit(specs[pi][0], specs[pi][1], x_all[:, : m + pi], y_np[:, pi], iters=iters)
            cpu_fit += seconds
            cpu_evals += evals
        cpu_scale = 1.0 if n <= 400 else float(p)
        rows.append({"n": n, "m": m, "p": p, "gpu_logpdf_ms": gpu[n][0], "cpu_logpdf_ms": 1e3 * cpu_logpdf,
                     "gpu_fit_ms": gpu[n][1], "gpu_fit_ms_all": gpu[n][2], "cpu_fit_ms": 1e3 * cpu_fit * cpu_scale,
                     "cpu_fit_sample": "every layer" if cpu_scale == 1.0 else f"the last layer x p = {p}",
                     "cpu_fit_evaluations": int(cpu_evals * cpu_scale)})
    return {"rows": rows, "cpu_cores": threads, "fit_iters": iters,
            "model": "m = 2, p = 4, linear output dependence (C2's model), complete data, normalise_y=False",
            "cpu_kind": "port: torch-CPU fp64 operators with autograd gradients, scipy L-BFGS-B over log-parameters (oracle/torch_cpu.py)"}


def lone_factorisation_leg(eng, sizes=(1024, 2048, 4096, 8192, 16384)):
    """gpar_potrf alone - what every layer of a `fit` and every rank of a layer-parallel evaluation runs - on the augmented
    (n + 1) x (n + 1) matrix of a log marginal likelihood: best of 5, milliseconds per size."""
    import torch

    from gpar_amd import hip

    out = {}
    for n in sizes:
        gen = torch.Generator(device="cpu")
        gen.manual_seed(n)
        pts = torch.rand(n, 4, generator=gen, dtype=torch.float64).to(eng.device)
        K0 = hip.alloc_matrix(n + 1, n + 1, eng.device, zero=True)
        K0[:n, :n] = torch.exp(-0.5 * torch.cdist(pts, pts) ** 2 / 0.25)
        K0[:n, :n].diagonal().add_(0.1)
        K0[n, :n] = torch.sin(5 * pts[:, 0])
        A = hip.alloc_matrix(n + 1, n + 1, eng.device)
        best = float("inf")
        for _ in range(5):
            A.copy_(K0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _, info = hip.potrf_(A, nf=n)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        out[str(n)] = best if int(info.item()) == 0 else None
        del K0, A, pts
        torch.cuda.empty_cache()
    return out


def cpu_baseline_leg(x_np, y_np, m, p, n_star=2048, num_samples=100, budget_s=30.0):
    """The reference's CPU path on this box's host cores, restated on the torch-CPU fp64 operators `lab.torch` dispatches
    to (oracle/torch_cpu.py: unfused Gram terms with expanded squared distances, torch.linalg.cholesky,
    solve_triangular; all host threads).  Layers of the SAME full-size workload are evaluated from the last (widest) one
    downwards until ~budget_s seconds of CPU work are spent; the per-layer mean is multiplied by p.  One posterior draw of
    one layer and one autograd objective + gradient evaluation are timed as well (the per-unit costs of `predict` / `fit`)."""
    import torch

    from gpar_amd.engine import set_engine
    from gpar_amd.regression import _construct_gpar
    from oracle import kernels as ok
    from oracle import torch_cpu as tc
    from oracle.engine import OracleEngine

    threads = tc.set_threads()
    previous = set_engine(OracleEngine())  # only to read the kernel specification off the host-side model objects
    try:
        reg = c3_regressor()
        gpar = _construct_gpar(reg, reg.vs, m, p)
        layers = []
        for pi in range(p):
            f, noise = gpar.layers[pi]()
            layers.append((ok.spec_to_dict(f.kernel.resolve(m + pi)), float(noise)))
    finally:
        set_engine(previous)
    n = x_np.shape[0]
    x_all = torch.as_tensor(np.concatenate([x_np, y_np], axis=1))
    measured, spent = [], 0.0
    last = None
    for pi in reversed(range(p)):
        spec, noise = layers[pi]
        design = x_all[:, : m + pi]
        value, stages, L = tc.layer_logpdf(spec, design, y_np[:, pi], np.full(n, noise))
        measured.append(dict(stages, layer=pi, logpdf=value))
        spent += sum(stages.values())
        if last is None:
            last = (spec, noise, design, L)
        if spent + spent / len(measured) > budget_s:  # the next layer would overrun the budget
            break
    per_layer = spent / len(measured)
    out = {
        "value": 1.0 / (per_layer * p),
        "unit": "logpdf/s",
        "cores": threads,
        "kind": "port",
        "backend": "torch-CPU fp64 (torch.linalg.cholesky / solve_triangular / exp / matmul: the operators the reference's "
                   "lab.torch backend dispatches to), one torch thread per CPU the container may use",
        "host_logical_cpus": os.cpu_count(),
        "cgroup_cpu_quota": tc.cpu_quota(),
        "sample": f"{len(measured)} of the p={p} layers of the same full-size workload (n={n}; from the last, widest layer "
                  f"downwards, {spent:.1f} s of CPU work), mean per layer x p; host has {os.cpu_count()} logical CPUs, "
                  f"the container's CPU quota is {tc.cpu_quota()}, {threads} torch threads",
        "per_layer_s": per_layer,
        "gram_s": float(np.mean([s["gram_s"] for s in measured])),
        "potrf_s": float(np.mean([s["potrf_s"] for s in measured])),
        "solve_s": float(np.mean([s["solve_s"] for s in measured])),
        "potrf_gflops": (n**3 / 3.0) / float(np.mean([s["potrf_s"] for s in measured])) * 1e-9,
        "layers": measured,
    }
    try:
        cpu = open("/proc/cpuinfo").read()
        out["cpu_model"] = next(line.split(":", 1)[1].strip() for line in cpu.splitlines() if line.startswith("model name"))
    except Exception:
        pass
    # per-unit costs of predict (one posterior draw of one layer) and fit (one objective + autograd gradient of one layer)
    spec, noise, design, L = last
    z = torch.linalg.solve_triangular(L, torch.as_tensor(y_np[:, p - 1]).reshape(-1, 1), upper=False)
    xs = torch.as_tensor(np.random.default_rng(5).uniform(0, 1, (n_star, design.shape[1])))
    _, stages = tc.layer_posterior_sample(spec, design, L, z, xs, np.full(n_star, noise))
    draw_s = sum(stages.values())
    out["predict"] = {
        "per_layer_per_sample_s": draw_s, "stages": stages, "n_star": n_star,
        "estimated_predict_s": p * per_layer + p * num_samples * draw_s,
        "note": f"reference predict = p conditionings (one factorisation each, as logpdf) + num_samples x p posterior draws "
                f"(cross-Gram, triangular solve against the n x n factor, n* x n* Cholesky): one draw of the last layer measured, "
                f"scaled to p={p} layers x {num_samples} samples",
    }
    if per_layer < 8.0:
        leaves, rebuild = tc.leaf_spec(spec)
        leaves.append(torch.tensor(noise, dtype=torch.float64))
        del L, z
        _, _, stages = tc.layer_objective_and_gradient(lambda ps: rebuild(ps[:-1]), leaves, design, y_np[:, p - 1], noise_index=-1)
        out["fit"] = {"objective_and_gradient_s": stages["forward_s"] + stages["backward_s"], "stages": stages,
                      "note": "one evaluation of one layer's training objective with torch autograd through Gram, Cholesky and "
                              "solve (what varz.minimise_l_bfgs_b calls once per L-BFGS-B function evaluation)"}
    return out


if __name__ == "__main__":
    main()
