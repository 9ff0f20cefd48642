"""MFMA-pipe utilisation of the factorisation kernels from hardware counters (passes collected by tools/pmc_mfma.sh):

    python tools/pmc_mfma.py gpurun_out/pmc_mfma > profiles/r01_bench_pmc_mfma.json

SQ_VALU_MFMA_BUSY_CYCLES sums, over the chip's 1024 SIMDs, the cycles a SIMD's matrix pipe was busy (checked against the
instruction count: a v_mfma_f64_16x16x4 holds it 64 cycles and the counter equals 64 x the number of such instructions the
launch issues); GRBM_GUI_ACTIVE is the busy-cycle count summed over the 8 XCDs.  Utilisation = busy / (GUI_ACTIVE / 8 x 1024),
a ratio of SUMS over the launches of a kernel (i.e. weighted by duration) - in cycles of whatever clock the chip ran at, so
it sits above achieved / peak-at-2.4-GHz whenever the clock is lower.  Counter collection serialises kernels: these are
figures for each kernel alone on the chip.  `MfmaUtil` is rocprofv3's derived metric (gfx94x formula), averaged per launch."""
import collections, csv, json, sys

root = sys.argv[1]


def table(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f"{root}/{sub}/pmc_counter_collection.csv")):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE, and --pmc MfmaUtil, "
                 "of `python bench.py --steps 1 --warmup 0 --no-extras --no-cpu` (tools/pmc_mfma.sh); `syrk`: the same counters on "
                 "tools/time_gemm.py 16384 512 1024 (bare K = 512 / 1024 trailing-update shape)",
       "kernels": {}}
raw, util = table("bench"), table("util")
for k, v in raw.items():
    if not any(s in k for s in ("gemm_f64_kernel", "potrf_panel_kernel")):
        continue
    busy, gui = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(v["GRBM_GUI_ACTIVE"])
    out["kernels"][k] = {
        "launches": len(v["GRBM_GUI_ACTIVE"]),
        "mfma_busy_simd_cycles": busy,
        "gui_active_cycles_per_xcd": gui / 8,
        "mfma_pipe_utilisation": busy / (gui / 8 * 1024),
        "MfmaUtil_mean_per_launch": sum(util[k]["MfmaUtil"]) / len(util[k]["MfmaUtil"]) if k in util else None,
    }
for k, v in table("syrk").items():
    if "gemm_f64_kernel" in k:
        busy, gui = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(v["GRBM_GUI_ACTIVE"])
        out["syrk"] = {"kernel": k, "launches": len(v["GRBM_GUI_ACTIVE"]), "mfma_pipe_utilisation": busy / (gui / 8 * 1024)}
print(json.dumps(out, indent=1))
