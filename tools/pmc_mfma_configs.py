"""Per-kernel matrix-pipe utilisation from the counter passes of tools/pmc_mfma_configs.sh:

    python tools/pmc_mfma_configs.py gpurun_out/pmc_cfg > profiles/r02_pmc_mfma.json

utilisation = SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip's 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs x 1024), a ratio of
sums over the launches of a kernel, in cycles of the clock the chip actually ran at; `share` = the kernel's share of the
summed GUI-active cycles of the run (kernels are serialised under counter collection)."""
import collections, csv, glob, json, sys

root = sys.argv[1]
out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --kernel-trace (tools/pmc_mfma_configs.sh)", "runs": {}}
for run in ("bench", "C2", "C4", "C5"):
    files = glob.glob(f"{root}/{run}/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            calls[k] += 1
    total = sum(v["GRBM_GUI_ACTIVE"] for v in agg.values())
    table = {}
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"]):
        if not k.startswith("gpar::"):
            continue
        gui = v["GRBM_GUI_ACTIVE"]
        table[k] = {"launches": calls[k], "share_of_active_cycles": round(gui / total, 4),
                    "mfma_pipe_utilisation": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * 1024), 4) if gui else None,
                    "mfma_f64_mops": v["SQ_INSTS_VALU_MFMA_MOPS_F64"]}
    out["runs"][run] = table
json.dump(out, sys.stdout, indent=1)
