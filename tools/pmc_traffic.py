"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --steps 1 --warmup 0 --no-extras --no-cpu`
into per-launch HBM traffic of the dominant kernel (the trailing SYRK update of gpar_potrf).

The trailing launches are identified by replaying gpar_potrf's host-side launch sequence (gpar_amd/csrc/potrf.h:
potrf_run / potrf_panel with the default policy) and matching it, in dispatch order, against the NT GEMM launches in
the counter CSV.  Corrections follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: counter values are KiB; on gfx950
FETCH_SIZE reports half the bytes of a wide coalesced streaming read, so the read side is doubled (an upper bound for
the 8-byte accesses of the epilogue, whose width is uncalibrated); WRITE_SIZE is used as reported.

    python tools/pmc_traffic.py gpurun_out/pmc_bench_FETCH_SIZE gpurun_out/pmc_bench_WRITE_SIZE N nf > profiles/..json
"""
import csv, json, sys


def launch_sequence(N, nf):
    """'T' for trailing-update GEMMs, 'I' for panel-internal ones, in launch order (default policy, look-ahead off)."""
    nbo = 512 if N >= 12288 else 256 if N >= 6144 else 128 if N >= 1536 else 64
    nbm = 128 if nbo >= 512 else 64
    seq, flops = [], []

    def update(k0, kend, col_end, Nloc, tag):
        rows, cols = Nloc - kend, col_end - kend
        if rows > 0 and cols > 0:
            seq.append(tag)
            flops.append(2.0 * (kend - k0) * (cols * (cols + 1) / 2 + (rows - cols) * cols))

    def panel(c0, c1, nb):
        w = c1 - c0
        if w <= 64:
            return
        if nb >= w:
            nb = nbm if (w > nbm and nbm >= 64) else 64
        nxt = nbm if nb > nbm else 64
        for k0 in range(c0, c1, nb):
            kend = min(k0 + nb, c1)
            panel(k0, kend, nxt)
            update(k0, kend, c1, N, "I")

    for k0 in range(0, nf, nbo):
        kend = min(k0 + nbo, nf)
        panel(k0, kend, nbo)
        if kend >= N:
            break
        update(k0, kend, N, N, "T")
    return seq, flops


def read(dirname, counter):
    rows = list(csv.DictReader(open(dirname + "/pmc_counter_collection.csv")))
    rows = [r for r in rows if r["Counter_Name"] == counter and "gemm_f64_kernel<false, true" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows]


def main():
    fdir, wdir, N, nf = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    seq, flops = launch_sequence(N, nf)
    fetch, write = read(fdir, "FETCH_SIZE"), read(wdir, "WRITE_SIZE")
    per = len(seq)
    assert len(fetch) % per == 0 and len(write) == len(fetch), (len(fetch), len(write), per)
    nfac = len(fetch) // per
    tf = [fetch[i] for i in range(len(fetch)) if seq[i % per] == "T"]
    tw = [write[i] for i in range(len(write)) if seq[i % per] == "T"]
    nT = len(tf)
    fetch_b = 2.0 * 1024.0 * sum(tf) / nT
    write_b = 1024.0 * sum(tw) / nT
    alg = sum(f for f, s in zip(flops, seq) if s == "T") / seq.count("T")
    # algorithmic C traffic of a lower-trapezoid read-modify-write: 16 bytes per stored element
    out = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 1 --warmup 0 --no-extras --no-cpu`",
        "kernel": "gemm_f64_kernel<false,true> trailing SYRK launches of gpar_potrf",
        "factorisations": nfac, "trailing_launches": nT,
        "fetch_bytes_per_launch_x2_corrected": fetch_b, "fetch_bytes_per_launch_raw": fetch_b / 2,
        "write_bytes_per_launch": write_b,
        "traffic_bytes_per_launch": fetch_b + write_b,
        "algorithmic_flops_per_launch": alg,
        "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (exact for 16-byte coalesced streams; the 8-byte epilogue reads are uncalibrated), WRITE_SIZE as reported",
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
