"""BASELINE.md section 4's results table from ONE bench.py line (+ the Gram timings of tools/time_gram_configs.py):
    python tools/baseline_table.py profiles/r06_bench_n1.json profiles/r06_gram_configs.jsonl > profiles/r06_baseline_table.md
Every number is read from those two files; cells the harness cannot fill on one GPU say why."""
import json, sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
gram = {}
if len(sys.argv) > 2:
    for line in open(sys.argv[2]):
        if line.strip():
            r = json.loads(line)
            gram[r["config"]] = r
g, fp, cb, rl = d["config_grid"], d["fit_predict"], d["cpu_baseline"], d["roofline"]
PEAK = rl["peak"]


def ms(v):
    return "-" if v is None else (f"{v / 1e3:.2f} s" if v >= 1e4 else f"{v:.1f}" if v >= 10 else f"{v:.2f}")


def tf(flops, millis):
    t = flops / (millis * 1e-3) * 1e-12
    return f"{t:.1f} TF / {100 * t / PEAK:.0f} %"


def gram_cell(c):
    r = gram.get(c)
    return "-" if r is None else f"{1e3 * r['tb_per_s_sustained']:.0f} GB/s / {100 * r['frac_of_hbm_peak_sustained']:.0f} %"


multi = "not measured (no multi-GPU node reached this run)"
parity = "logpdf 1e-10 vs oracle, moments rtol 1e-8 (tests); reference pin: " + str(d.get("parity_pin"))
rows = []
c2, c2c = g["C2"], g["C2"].get("cpu_baseline", {})
fit_cpu = None if "fit_evaluation_ms" not in c2c else c2c["fit_evaluation_ms"] * c2.get("fit_evaluations", 0)
rows.append(("C2", "logpdf / fit(20) / predict(100)",
             f"{ms(c2c.get('logpdf_ms'))} / {ms(fit_cpu)} (est.) / {ms(c2c.get('predict_ms'))} (est.) ({c2c.get('cores')})",
             f"{ms(c2['logpdf_ms_best'])} / {ms(c2['fit_20_iters_ms'])} / {ms(c2['predict_100_samples_ms'])}",
             tf(c2["algorithmic_flops"], c2["logpdf_ms_best"]), gram_cell("C2")))
c3_fit_cpu = None if "fit" not in cb else 1e3 * cb["fit"]["objective_and_gradient_s"] * fp["fit_evaluations"]
c3_pred_cpu = None if "predict" not in cb else 1e3 * cb["predict"]["estimated_predict_s"]
rows.append(("C3", "logpdf / fit(20) / predict(100)",
             f"{ms(1e3 / cb['value'])} / {ms(c3_fit_cpu)} (est.) / {ms(c3_pred_cpu)} (est.) ({cb['cores']})",
             f"{ms(d['ms_per_step'])} / {ms(fp['fit_ms'])} / {ms(fp['predict_ms'])}",
             f"{rl['achieved']:.1f} TF / {100 * rl['frac']:.0f} % (update kernel, live); step {tf(d['config'].get('algorithmic_flops', 8 * (16384 ** 3 / 3 + 16384 ** 2)), d['ms_per_step'])}",
             gram_cell("C3")))
c4, c4c = g["C4"], g["C4"].get("cpu_baseline", {})
rows.append(("C4", "logpdf / condition / predict(100)",
             f"{ms(c4c.get('logpdf_ms'))} / {ms(c4c.get('condition_ms'))} / {ms(c4c.get('predict_ms'))} (est.) ({c4c.get('cores')})",
             f"{ms(c4['logpdf_ms_best'])} / {ms(c4.get('condition_ms'))} / {ms(c4.get('predict_ms'))}",
             tf(c4["algorithmic_flops"], c4["logpdf_ms_best"]), gram_cell("C4")))
c5, c5c = g["C5"], g["C5"].get("cpu_baseline", {})
rows.append(("C5", "logpdf / predict(S=200)",
             f"{ms(c5c.get('logpdf_ms'))} / {ms(c5c.get('predict_ms'))} (est.) ({c5c.get('cores')})",
             f"{ms(c5['logpdf_ms_best'])} / {ms(c5['predict_200_samples_ms'])}",
             tf(c5["algorithmic_flops"], c5["logpdf_ms_best"]), gram_cell("C5")))
c1, c1c = g["C1"], g["C1"].get("cpu_baseline", {})
rows.append(("C1", "logpdf / fit + predict(200) as examples/paper/synthetic.py",
             f"{ms(c1c.get('logpdf_ms'))} / {ms(c1c.get('fit_predict_ms'))} ({c1c.get('cores')})",
             f"{ms(c1['logpdf_ms_best'])} / {ms(c1.get('fit_predict_ms'))}", "(plumbing config: 25 points)", "-"))
print(f"BASELINE.md section 4, filled from `{sys.argv[1]}` (one MI355X, fp64; CPU = torch-CPU port on `cores` host cores, `{cb.get('cpu_model', '?')}`;")
print("(est.) = one timed unit of the same full-size workload scaled as BASELINE.md section 4 allows; ms unless marked s)\n")
print("| config | op | CPU ms (cores) | 1 x MI355X ms | 2 / 4 / 8 GPUs | achieved on the algorithmic flops of logpdf / % of 78.6 TF | Gram GB/s / % of 8 TB/s | parity |")
print("|---|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| {r[0]} | {r[1]} | {r[2]} | {r[3]} | {multi} | {r[4]} | {r[5]} | {parity} |")
