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
  cpu_baseline  the CPU oracle ("port" of the reference's torch-CPU/LAPACK path: numpy Gram + LAPACK potrf) timed
                on this box's host cores on a bounded sample of the same workload;
  fit_predict   wall-clock of a short `fit` + `predict` at the same size (the other half of BASELINE's metric).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

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


def c3_regressor():
    from gpar_amd.regression import GPARRegressor

    return GPARRegressor(scale=0.5, linear=True, nonlinear=True, markov=2, noise=0.1, replace=False, impute=True,
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
    ap.add_argument("--extras-timeout", type=float, default=600.0,
                    help="seconds the fit + predict / CPU-baseline / teardown part may take before every rank exits (rank 0 prints the line first)")
    ap.add_argument("--cpu-n", type=int, default=0, help="rows of the bounded CPU sample (0: all n rows, ~17 s at C3)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
        if one_gpu_dev:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

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

    def step():
        return sharded_logpdf(gpar, x, y, w)

    value = None
    for _ in range(args.warmup):
        value = step()
    lib.gpar_profile_read(None, None, None, None, 1)
    lib.gpar_profile_enable(1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        value = step()
    barrier()
    elapsed = time.perf_counter() - t0
    lib.gpar_profile_enable(0)
    import ctypes

    launches, ms, busy, flops = ctypes.c_int(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    lib.gpar_profile_read(ctypes.byref(launches), ctypes.byref(ms), ctypes.byref(busy), ctypes.byref(flops), 1)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=eng.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

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
            "parallelism": f"layer-parallel x{world} (layer i on rank i mod {world}; 8-byte all-reduce only)",
            "logpdf": float(value),
        },
    }
    if launches.value > 0 and busy.value > 0:
        # Independent layers are pipelined over two streams, so two trailing updates (of different layers) are often in
        # flight at once and each one's own duration covers work of both.  The kernel's rate is therefore taken over the
        # UNION of the launch intervals: flops / busy time = per-launch flops / (average launch duration / concurrency).
        achieved = flops.value / (busy.value * 1e-3) * 1e-12
        out["roofline"] = {
            "kernel": "gpar::gemm_f64_kernel<false, true, 1, 128> and its half-tile form <false, true, 1, 64> for launches of at most 256 tiles (every trailing-update launch of gpar_potrf: rank-512 / rank-1024 SYRK of the rest of the matrix and the narrow look-ahead slices; v_mfma_f64_16x16x4)",
            "bound": "mfma",
            "achieved": achieved,
            "peak": FP64_MATRIX_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / FP64_MATRIX_PEAK_TFLOPS,
            "traffic": pmc_traffic(n, m, p),
            "traffic_source": "profiles/r01_bench_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; bytes per launch)",
            "launches": launches.value,
            "flop_per_launch": flops.value / launches.value,
            "avg_launch_ms": ms.value / launches.value,
            "concurrency": ms.value / busy.value,
            "busy_ms": busy.value,
            "rank": 0,
        }

    # The timed region overlaps the trailing SYRK of panel k with the fused factorisation of panel k+1 (look-ahead on a
    # second stream) and with the other stream's layer, so the live number above is the kernel's rate WHILE SHARING the
    # chip.  One extra, untimed evaluation with look-ahead and layer pipelining switched off gives the same kernel's
    # per-launch rate when it has the GPU to itself (concurrency 1: flops per launch / average launch duration).
    if True:  # every rank takes part (the evaluation contains a collective), whether or not it owns a layer
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
                                                       "GPAR_LAYER_PIPELINE=0 (nothing co-running): flops per launch / average launch duration"}
        finally:
            del os.environ["GPAR_POTRF_LOOKAHEAD"]
            del os.environ["GPAR_LAYER_PIPELINE"]
        if "roofline" in out:
            out["roofline"]["note"] = ("live value, measured inside the timed region with hipEvents on the launch streams: flops of all "
                                       "launches / union of their intervals (`concurrency` launches are in flight on average because "
                                       "independent layers run on two streams, and each co-runs with the next panel's factorisation); "
                                       "`avg_launch_ms` is the plain per-launch average that `rocprofv3 --stats` reports; see `isolated` "
                                       "for the kernel alone")

    # The headline is measured by now.  Everything below (the fit + predict leg with its collectives, the CPU baseline, the
    # process-group teardown) runs under a watchdog: should a peer die or a collective hang, every rank leaves after
    # `--extras-timeout` seconds and rank 0 still prints the one JSON line, with the leg marked as timed out.
    import threading

    printed = []

    def emit():
        if rank == 0 and not printed:
            printed.append(True)
            print(json.dumps(out), flush=True)

    def bail():
        if not args.no_extras:
            out.setdefault("fit_predict", {"error": f"timed out after {args.extras_timeout} s", "n_gpus": world})
        emit()
        os._exit(0)

    watchdog = threading.Timer(args.extras_timeout, bail)
    watchdog.daemon = True
    watchdog.start()
    if not args.no_extras:  # every rank takes part (sharded fit / predict contain collectives)
        try:
            leg = fit_predict_leg(eng, x_np, y_np, n, m, p, world)
        except Exception as exc:  # the headline line must survive a failure of the extra leg
            leg = {"error": f"{type(exc).__name__}: {exc}", "n_gpus": world}
        if rank == 0:
            out["fit_predict"] = leg
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline_leg(x_np, y_np, m, p, args.cpu_n if 0 < args.cpu_n < n else n, n)
    emit()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    watchdog.cancel()


def pmc_traffic(n, m, p):
    """HBM bytes per trailing-SYRK launch from the committed PMC passes of this very workload (null for any other
    workload: counters cannot be collected from inside the timed process)."""
    path = os.path.join(ROOT, "profiles", "r01_bench_pmc_traffic.json")
    if (n, m, p) != (16384, 4, 8) or not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get("traffic_bytes_per_launch")


def fit_predict_leg(eng, x_np, y_np, n, m, p, world, fit_iters=2, num_samples=8, n_star=1024):
    """Short fit (fixed L-BFGS-B iteration count) + predict on the same data - the other half of BASELINE.json's
    metric.  One rank: GPARRegressor.fit / predict.  Several ranks: layer pi is trained on rank pi mod N
    (parallel.sharded_fit, hyper-parameters broadcast afterwards) and the posterior samples are split over the ranks
    (parallel.sharded_sample; every rank conditions all layers, so the conditioning part does not scale)."""
    import torch

    from gpar_amd.parallel import sharded_fit, sharded_sample

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
            torch.cuda.synchronize()

    reg = c3_regressor()
    xs = np.random.default_rng(5).uniform(0, 1, (n_star, m))
    sync()
    t0 = time.perf_counter()
    if world == 1:
        reg.fit(x_np, y_np, iters=fit_iters)
    else:
        sharded_fit(reg, x_np, y_np, iters=fit_iters)
    sync()
    t1 = time.perf_counter()
    if world == 1:
        mean = reg.predict(xs, num_samples=num_samples, latent=True)
    else:
        mean = np.mean(sharded_sample(reg, xs, num_samples=num_samples, latent=True), axis=0)
    sync()
    t2 = time.perf_counter()
    return {"fit_ms": 1e3 * (t1 - t0), "fit_iters": fit_iters, "predict_ms": 1e3 * (t2 - t1), "num_samples": num_samples,
            "n_star": n_star, "predict_mean_abs": float(np.mean(np.abs(mean))), "n_gpus": world,
            "timing": "barrier-bracketed wall-clock on rank 0"}


def cpu_baseline_leg(x_np, y_np, m, p, n_sub, n_full):
    """The CPU oracle (numpy Gram + LAPACK Cholesky; a port of the reference's torch-CPU path) on ONE layer of the
    workload (the last, widest design matrix) - at full size by default, ~17 s on the GPU box's host - times p."""
    from gpar_amd.engine import set_engine
    from gpar_amd.regression import _construct_gpar
    from oracle.engine import OracleEngine

    try:
        from threadpoolctl import threadpool_info

        threads = max([int(i.get("num_threads", 1)) for i in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    previous = set_engine(OracleEngine())
    try:
        reg = c3_regressor()
        gpar = _construct_gpar(reg, reg.vs, m, p)
        xs, ys = x_np[:n_sub], y_np[:n_sub]
        design = np.concatenate([xs, ys[:, : p - 1]], axis=1)
        f, noise = gpar.layers[p - 1]()
        t0 = time.perf_counter()
        val = float(f(design, float(noise)).logpdf(ys[:, p - 1]))
        dt = time.perf_counter() - t0
    finally:
        set_engine(previous)
    per_layer_full = dt * (n_full / n_sub) ** 3
    how = ("no extrapolation in n" if n_sub == n_full else
           f"extrapolated to n={n_full} with the n^3 law (pessimistic for the CPU: the Gram build is n^2)")
    return {
        "value": 1.0 / (per_layer_full * p),
        "unit": "logpdf/s",
        "cores": threads,
        "kind": "port",
        "sample": f"one of the p={p} layers (the last, widest) of the same model on {n_sub} of {n_full} rows: {dt:.2f} s measured "
                  f"(numpy fused-by-term Gram + LAPACK dpotrf via numpy/scipy, {threads} BLAS threads, host has "
                  f"{os.cpu_count()} logical CPUs); {how}; multiplied by p={p} near-equal layers",
        "measured_s": dt,
        "sample_logpdf": val,
    }


if __name__ == "__main__":
    main()
