"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --steps 1 --warmup 0 --no-extras --no-cpu`
into per-launch HBM traffic of the dominant kernel: the trailing SYRK update of gpar_potrf, which has its own kernel
symbols (gemm_f64_kernel<false, true, 1, 128> and, for launches with at most 256 tiles, its half-tile form <false, true, 1, 64>).

Corrections follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: counter values are KiB; on gfx950 FETCH_SIZE reports
half the bytes of a wide coalesced streaming read, so the read side is doubled (exact for the 16-byte operand
streams and, since the epilogue transposes through LDS, for the 16-byte read-modify-write of C); WRITE_SIZE as reported.

    python tools/pmc_traffic.py gpurun_out/pmc_bench_FETCH_SIZE gpurun_out/pmc_bench_WRITE_SIZE > profiles/r01_bench_pmc_traffic.json
"""
import csv, json, sys

KERNEL = "gemm_f64_kernel<false, true, 1,"  # both tile forms: every trailing-update launch of gpar_potrf, look-ahead slices included


def read(dirname, counter):
    import glob

    path = sorted(glob.glob(dirname + "/**/*counter_collection.csv", recursive=True))[0]
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == counter and KERNEL in r["Kernel_Name"]]


def main():
    fetch, write = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE")
    assert len(fetch) == len(write) and fetch, (len(fetch), len(write))
    # the command runs ONE timed evaluation (lock-step: batched launches) and then bench.py's untimed `isolated` evaluation (layer
    # after layer: `iso` launches, argv[3]); the per-launch figure is taken over the timed evaluation's launches only, like
    # roofline.achieved / flop_per_launch
    iso = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    if 0 < iso < len(fetch):
        fetch, write = fetch[:-iso], write[:-iso]
    n = len(fetch)
    fetch_b = 2.0 * 1024.0 * sum(fetch) / n
    write_b = 1024.0 * sum(write) / n
    print(json.dumps({
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 1 --warmup 0 --no-extras --no-cpu`",
        "kernel": "gpar::gemm_f64_kernel<false, true, 1, 128> + <false, true, 1, 64> (all trailing-update launches of the timed, lock-step evaluation: gpar_potrf_batch)",
        "launches": n,
        "gemm_source_sha16": __import__("hashlib").sha256(open(__import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "gpar_amd", "csrc", "gemm_f64.h"), "rb").read()).hexdigest()[:16],
        "fetch_bytes_per_launch_x2_corrected": fetch_b, "fetch_bytes_per_launch_raw": fetch_b / 2,
        "write_bytes_per_launch": write_b,
        "traffic_bytes_per_launch": fetch_b + write_b,
        "traffic_bytes_per_evaluation": (fetch_b + write_b) * n,
        "isolated_launches_excluded": iso,
        "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (calibrated for 16-byte coalesced streams, which operands and C now both are), WRITE_SIZE as reported",
    }, indent=1))


if __name__ == "__main__":
    main()
