"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
    python bench.py --steps 1 --warmup 0 --no-extras --no-cpu --no-isolated
into the HBM traffic of the dominant kernel: the trailing SYRK update of gpar_potrf, which has its own kernel symbols
(gemm_f64_kernel<false, true, 1, 128> and, for launches with at most 256 tiles, its half-tile form <false, true, 1, 64>).

The command runs ONE lock-step evaluation and nothing else (`--no-isolated`: bench.py's second, layer-after-layer evaluation is
left out), so EVERY dispatch of the named kernel in the trace belongs to that evaluation - nothing is sliced by a count taken
elsewhere.  (Round 5 dropped "the last N" dispatches with N = the in-library hook's event count, which also counted the two
one-wave updates of the augmented row per factorisation: the mean was taken over the 25 largest of 41 launches.)

Corrections follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: counter values are KiB; on gfx950 FETCH_SIZE reports
half the bytes of a wide coalesced streaming read, so the read side is doubled (exact for the 16-byte operand
streams and for the 8-byte full-line loads of C into the accumulators); WRITE_SIZE as reported.

    python tools/pmc_traffic.py gpurun_out/refresh/pmc_FETCH_SIZE gpurun_out/refresh/pmc_WRITE_SIZE > profiles/r06_bench_pmc_traffic.json
"""
import csv, glob, hashlib, json, os, sys

KERNEL = "gemm_f64_kernel<false, true, 1,"  # both tile forms: every trailing-update launch of gpar_potrf, look-ahead slices included
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCHEDULE_SOURCES = ("gemm_f64.h", "potrf.h", "panel2.h")   # the kernel AND the launch schedule (which launches exist, how big)


def schedule_sha16():
    h = hashlib.sha256()
    for name in SCHEDULE_SOURCES:
        with open(os.path.join(ROOT, "gpar_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def read(dirname, counter):
    path = sorted(glob.glob(dirname + "/**/*counter_collection.csv", recursive=True))[0]
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"]))
    # (one row per dispatch and counter; a counter sampled per XCD / shader engine comes as several rows of one dispatch: summed)
    per = {}
    for r in rows:
        if r["Counter_Name"] == counter and KERNEL in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])] = per.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return [per[k] for k in sorted(per)], sorted(per)


def main():
    (fetch, ids_f), (write, ids_w) = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE")
    assert len(fetch) == len(write) and fetch, (len(fetch), len(write))
    n = len(fetch)
    fetch_eval = 2.0 * 1024.0 * sum(fetch)
    write_eval = 1024.0 * sum(write)
    print(json.dumps({
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 1 --warmup 0 --no-extras --no-cpu --no-isolated`",
        "kernel": "gpar::gemm_f64_kernel<false, true, 1, 128> + <false, true, 1, 64>: ALL dispatches of the trace = the trailing-update launches of the one lock-step evaluation (gpar_potrf_batch)",
        "launches": n,
        "dispatch_ids": [ids_f[0], ids_f[-1]],
        "schedule_source_sha16": schedule_sha16(),
        "schedule_sources": list(SCHEDULE_SOURCES),
        "fetch_bytes_per_evaluation_x2_corrected": fetch_eval, "fetch_bytes_per_evaluation_raw": fetch_eval / 2,
        "write_bytes_per_evaluation": write_eval,
        "traffic_bytes_per_evaluation": fetch_eval + write_eval,
        "traffic_bytes_per_launch": (fetch_eval + write_eval) / n,
        "fetch_bytes_per_launch_x2_corrected": fetch_eval / n,
        "write_bytes_per_launch": write_eval / n,
        "per_launch_bytes": [round(2048.0 * f + 1024.0 * w) for f, w in zip(fetch, write)],
        "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (calibrated for wide coalesced streams, which operands and C both are), WRITE_SIZE as reported; "
                "mean over ALL launches of the evaluation",
    }, indent=1))


if __name__ == "__main__":
    main()
