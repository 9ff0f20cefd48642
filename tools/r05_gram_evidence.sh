# Round 5 Gram evidence (run through gpurun): timings of the final kernels on every BASELINE configuration, the lock-step build of
# C2 from a kernel trace, and HBM counters (separate --pmc passes: FETCH_SIZE, WRITE_SIZE) for the C3 and C5 builds.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_gram; rm -rf $O; mkdir -p $O
python tools/time_gram_configs.py C2 C3 C4 C5 2>/dev/null > $O/gram_configs.jsonl
GPAR_GRAM_JIT_MIN_ENTRIES=-1 python tools/time_gram_configs.py C5 2>/dev/null | sed 's/"config": "C5"/"config": "C5 (interpreter)"/' >> $O/gram_configs.jsonl
for c in FETCH_SIZE WRITE_SIZE; do
    for cfg in C3 C5; do
        rocprofv3 --pmc $c --kernel-trace -f csv -d $O/pmc_${cfg}_$c -o pmc -- python tools/time_gram_configs.py $cfg > $O/pmc_${cfg}_$c.log 2>&1
    done
done
rocprofv3 --kernel-trace --stats -f csv -d $O/c2 -o kt -- python tools/run_config.py C2 --evals 6 --warmup 2 > $O/c2.log 2>&1
python - <<'PY' > gpurun_out/r5_gram/summary.json
import csv, glob, json
out = {}
for cfg, n in (("C3", 16384), ("C5", 8192)):
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = []
        for f in glob.glob(f"gpurun_out/r5_gram/pmc_{cfg}_{c}/**/*counter_collection.csv", recursive=True):
            rows += [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and ("gram_jit" in r["Kernel_Name"] or "gram_kernel" in r["Kernel_Name"])]
        vals[c] = sum(rows) / max(len(rows), 1)
        vals[c + "_launches"] = len(rows)
    alg = 8.0 * n * (n + 1) / 2
    fetch_b, write_b = 2.0 * 1024.0 * vals["FETCH_SIZE"], 1024.0 * vals["WRITE_SIZE"]   # KB units; FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM / rocprofv3 section)
    out[cfg] = {"algorithmic_bytes": alg, "fetch_bytes_x2_corrected": fetch_b, "write_bytes": write_b, "traffic_over_algorithmic": (fetch_b + write_b) / alg,
                "launches": vals["FETCH_SIZE_launches"]}
rows = list(csv.DictReader(open(glob.glob("gpurun_out/r5_gram/c2/**/*kernel_trace.csv", recursive=True)[0])))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in rows if "lockstep_build_kernel" in r["Kernel_Name"]]
d = d[2:] if len(d) > 2 else d
n, p = 4096, 4
bytes_build = p * 8.0 * (n + 1) * (n + 2) / 2
out["C2_lockstep_build_kernel"] = {"launches": len(d), "avg_us": sum(d) / len(d), "algorithmic_bytes": bytes_build,
                                   "tb_per_s": bytes_build / (sum(d) / len(d) * 1e-6) * 1e-12, "frac_of_hbm_peak": bytes_build / (sum(d) / len(d) * 1e-6) * 1e-12 / 8.0,
                                   "note": "one launch builds the four augmented layer matrices of C2 (lower triangles + observation rows)"}
print(json.dumps(out, indent=1))
PY
cat gpurun_out/r5_gram/summary.json
find $O -name "*.db" -delete
