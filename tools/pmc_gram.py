"""HBM traffic and rate of the Gram kernel (the memory-side kernel of the path) from the same two rocprofv3 --pmc passes
of `bench.py --steps 1 --warmup 0 --no-extras --no-cpu` that tools/pmc_traffic.py uses, plus its duration from a
plain kernel trace (the counter passes serialise kernels, which is what an isolated duration needs).

    python tools/pmc_gram.py gpurun_out/pmc_bench_FETCH_SIZE gpurun_out/pmc_bench_WRITE_SIZE > profiles/r01_gram_hbm.json
"""
import csv, json, sys


def counters(dirname, counter):
    rows = list(csv.DictReader(open(dirname + "/pmc_counter_collection.csv")))
    return [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == counter and "gram_kernel" in r["Kernel_Name"]]


def durations(dirname):
    rows = list(csv.DictReader(open(dirname + "/pmc_kernel_trace.csv")))
    return [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9 for r in rows if "gram_kernel" in r["Kernel_Name"]]


fetch, write, dur = counters(sys.argv[1], "FETCH_SIZE"), counters(sys.argv[2], "WRITE_SIZE"), durations(sys.argv[2])
n = 16384
# layers 1..7 of C3 (the first layer has no output-kernel terms and is cheaper): take the launches of the widest kernels
idx = [i for i, d in enumerate(dur) if d > 0.6 * max(dur)]
fb = 2.0 * 1024.0 * sum(fetch[i] for i in idx) / len(idx)
wb = 1024.0 * sum(write[i] for i in idx) / len(idx)
t = sum(dur[i] for i in idx) / len(idx)
algorithmic = 8.0 * n * (n + 1) / 2
print(json.dumps({
    "kernel": "gpar::gram_kernel (fused composite-kernel Gram build, lower triangle + noise diagonal + jitter), C3 layers 1-7",
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `python bench.py --steps 1 --warmup 0 --no-extras --no-cpu`",
    "launches_averaged": len(idx),
    "algorithmic_bytes_per_launch": algorithmic,
    "write_bytes_per_launch": wb, "fetch_bytes_per_launch_x2_corrected": fb,
    "traffic_over_algorithmic": (wb + fb) / algorithmic,
    "duration_ms": 1e3 * t,
    "achieved_GBps": (wb + fb) / t * 1e-9, "hbm_peak_GBps": 8000.0, "frac_of_hbm_peak": (wb + fb) / t / 8e12,
    "note": "every entry is written exactly once and nothing n x n is read; the fetches are the 64-row feature panels of the tiles "
            "(L2 misses of a 2 MB array that every tile re-reads).  The kernel is bound by the vector ALUs (generic evaluation of a "
            "run-time kernel specification + two fp64 exponentials per entry), not by HBM; it is ~1 % of the step",
}, indent=1))
