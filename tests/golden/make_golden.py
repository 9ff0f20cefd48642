"""Generates tests/golden/gpar_cases.json: small, fully specified GPAR problems with their expected values.

Expected values come from the CPU oracle's closed-form route (oracle/gp_ref.py, oracle/gpar_ref.py: slogdet +
solve, no Cholesky), evaluated in fp64 in the build container.  The reference implementation itself cannot be
imported (its stheno / lab / matrix / varz dependencies are not installable offline, see oracle/__init__.py),
so these vectors pin the *restated* algorithm; they travel to the GPU box as plain data.

    python tests/golden/make_golden.py          # rewrites gpar_cases.json deterministically
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import gp_ref, gpar_ref  # noqa: E402


def hypers_for(m, p, config, rng):
    """Hyper-parameter dictionary with the reference's names, random values."""
    h = {}
    for pi in range(p):
        _, p_inds = gpar_ref._indices(m, pi, config.get("markov"))
        pn = len(p_inds)
        h[f"{pi}/input/var"] = rng.uniform(0.5, 2.0)
        if not (config.get("scale_tie") and pi > 0):
            h[f"{pi}/input/scales"] = rng.uniform(0.5, 2.0, m).tolist()
        if config.get("rq"):
            h[f"{pi}/input/alpha"] = rng.uniform(0.3, 3.0)
        if config.get("per"):
            h[f"{pi}/input/per/var"] = rng.uniform(0.5, 2.0)
            h[f"{pi}/input/per/scales"] = rng.uniform(0.5, 2.0, 2 * m).tolist()
            h[f"{pi}/input/per/pers"] = rng.uniform(0.7, 1.5, m).tolist()
            h[f"{pi}/input/per/decay"] = rng.uniform(3.0, 10.0, m).tolist()
        if config.get("input_linear"):
            h[f"{pi}/input/lin/scales"] = rng.uniform(1.0, 5.0, m).tolist()
            h[f"{pi}/input/lin/const"] = rng.uniform(0.1, 1.0)
        if config.get("linear", True) and pi > 0:
            h[f"{pi}/output/lin/scales"] = rng.uniform(1.0, 5.0, pn).tolist()
        if config.get("nonlinear") and pi > 0:
            h[f"{pi}/output/nonlin/var"] = rng.uniform(0.5, 2.0)
            h[f"{pi}/output/nonlin/scales"] = rng.uniform(0.5, 2.0, pn).tolist()
            if config.get("rq"):
                h[f"{pi}/output/nonlin/alpha"] = rng.uniform(0.3, 3.0)
        h[f"{pi}/noise"] = rng.uniform(0.02, 0.2)
    return h


CASES = [
    ("eq-linear-2out", dict(n=12, m=1, p=2, config=dict(linear=True), impute=False, replace=False, missing=0.0, weights=False)),
    ("paper-synthetic-shape", dict(n=25, m=1, p=3, config=dict(linear=True, nonlinear=True), impute=True, replace=False, missing=0.0, weights=False)),
    ("weights-2in", dict(n=16, m=2, p=3, config=dict(linear=True, nonlinear=True), impute=False, replace=False, missing=0.0, weights=True)),
    ("markov2-4out", dict(n=14, m=2, p=4, config=dict(linear=True, nonlinear=True, markov=2), impute=False, replace=False, missing=0.0, weights=True)),
    ("markov0-igp", dict(n=10, m=1, p=3, config=dict(linear=True, nonlinear=True, markov=0), impute=False, replace=False, missing=0.0, weights=False)),
    ("rq-per-inputlinear", dict(n=18, m=2, p=2, config=dict(linear=True, nonlinear=True, rq=True, per=True, input_linear=True), impute=False, replace=False, missing=0.0, weights=True)),
    ("scale-tie", dict(n=11, m=2, p=3, config=dict(linear=True, scale_tie=True), impute=False, replace=False, missing=0.0, weights=False)),
    ("missing-impute", dict(n=20, m=1, p=3, config=dict(linear=True, nonlinear=True), impute=True, replace=False, missing=0.25, weights=True)),
    ("missing-noimpute", dict(n=20, m=1, p=3, config=dict(linear=True, nonlinear=True), impute=False, replace=False, missing=0.25, weights=False)),
    ("replace", dict(n=15, m=1, p=3, config=dict(linear=True, nonlinear=True), impute=True, replace=True, missing=0.15, weights=False)),
]

# Shapes of the reference's example workloads (synthetic stand-ins: the datasets are downloaded at run time there):
#   air_temp.py:18,27-46   inducing points on an even grid, replace + impute, B.epsilon = 1e-6, D-GPAR-L-NL
#   eeg.py:21-30           a block of the last outputs missing on the last rows, nonlinear only, noise 0.01
#   exchange.py:21-32      RQ kernels, linear + nonlinear, three outputs with missing blocks in the middle
WORKLOADS = [
    ("air-temp-shape", dict(n=30, p=3, x_ind=9, epsilon=1e-6, impute=True, replace=True, pattern="scattered",
                            config=dict(linear=True, nonlinear=True))),
    ("eeg-shape", dict(n=28, p=4, x_ind=None, epsilon=1e-12, impute=True, replace=False, pattern="tail-block",
                       config=dict(linear=False, nonlinear=True))),
    ("exchange-shape", dict(n=26, p=3, x_ind=None, epsilon=1e-12, impute=True, replace=False, pattern="middle-blocks",
                            config=dict(linear=True, nonlinear=True, rq=True))),
]


# Micro-cases: ONE recalled detail of the third-party arithmetic each (the items marked "M": recalled, not
# readable in /root/reference), with values chosen so that the alternative reading changes the number far beyond any tolerance.
# One run of make_reference_golden.py then settles each of them individually.
#   per-sincos-scales       m = 1: the two scales of the periodic embedding [sin, cos] differ by 6x (which is sin's?)
#   per-block-order         m = 2: per/scales = [s1, s2, c1, c2] - sin block then cos block, or interleaved per input?
#   per-decay-and-period    the locally periodic term: EQ(embedding / per_scale) * EQ(x / decay), period T in 2 pi x / T
#   rq-small-alpha          RQ = (1 + r^2 / (2 alpha))^-alpha at the reference's initial alpha = 1e-2
#   input-linear-const      (x / s)(x' / s)^T + c with the unbounded additive constant
#   weights-noise-over-w    noise / w with weights from 0.1 to 10 (not noise * w, not noise / w^2)
#   jitter-visible          B.epsilon = 1e-3 next to a noise of 1e-6: where the jitter is added, and that it is added once
#   markov1-window          markov = 1: layer i sees output i - 1 only
#   normalise-quirk         normalise_y=True: logpdf applies the UN-normalising map to its argument (gpar/regression.py:483, sic), with
#                           mean / population standard deviation (ddof = 0) of the non-missing training outputs
MICRO = [
    ("micro-per-sincos-scales", dict(n=9, m=1, p=1, config=dict(linear=False, per=True), set={"0/input/per/scales": [0.4, 2.4]})),
    ("micro-per-block-order", dict(n=10, m=2, p=1, config=dict(linear=False, per=True), set={"0/input/per/scales": [0.3, 0.9, 1.8, 3.6]})),
    ("micro-per-decay-and-period", dict(n=10, m=1, p=1, config=dict(linear=False, per=True),
                                        set={"0/input/per/pers": [0.37], "0/input/per/decay": [0.8], "0/input/per/var": 3.0, "0/input/var": 0.05})),
    ("micro-rq-small-alpha", dict(n=10, m=2, p=2, config=dict(linear=True, nonlinear=True, rq=True),
                                  set={"0/input/alpha": 1e-2, "1/input/alpha": 1e-2, "1/output/nonlin/alpha": 1e-2})),
    ("micro-input-linear-const", dict(n=9, m=2, p=1, config=dict(linear=False, input_linear=True),
                                      set={"0/input/lin/const": 0.35, "0/input/lin/scales": [0.7, 2.0], "0/input/var": 0.1})),
    ("micro-weights-noise-over-w", dict(n=12, m=1, p=1, config=dict(linear=False), weights=(0.1, 10.0), set={"0/noise": 0.5})),
    ("micro-jitter-visible", dict(n=10, m=1, p=2, config=dict(linear=True, nonlinear=True), epsilon=1e-3, set={"0/noise": 1e-6, "1/noise": 1e-6})),
    ("micro-markov1-window", dict(n=10, m=1, p=3, config=dict(linear=True, nonlinear=True, markov=1))),
    ("micro-normalise-quirk", dict(n=8, m=1, p=2, config=dict(linear=True, nonlinear=True), train=11)),
]


def micro_case(name, c, rng):
    n, m, p = c["n"], c["m"], c["p"]
    x = rng.uniform(-1.0, 1.0, (n, m))
    y = rng.standard_normal((n, p))
    hypers = hypers_for(m, p, c["config"], rng)
    for key, value in c.get("set", {}).items():
        assert key in hypers, key
        hypers[key] = value
    w = None
    if "weights" in c:
        lo, hi = c["weights"]
        w = np.exp(rng.uniform(np.log(lo), np.log(hi), (n, p)))
    eps = c.get("epsilon", 1e-12)
    out = {"name": name, "config": c["config"], "impute": False, "replace": False, "epsilon": eps, "x_ind": None, "x": x.tolist(),
           "y": y.tolist(), "w": None if w is None else w.tolist(), "hypers": hypers}
    y_eval = y
    if "train" in c:
        # the regressor is conditioned on (train_x, train_y) with normalise_y=True and then asked for the PRIOR logpdf of (x, y):
        # the reference maps y through _unnormalise_y first (sic), i.e. evaluates y * std + mean
        tx = rng.uniform(-1.0, 1.0, (c["train"], m))
        ty = 3.0 + 2.5 * rng.standard_normal((c["train"], p))
        ty[2, 1] = np.nan
        mean = np.array([np.mean(ty[~np.isnan(ty[:, i]), i]) for i in range(p)])
        std = np.array([np.std(ty[~np.isnan(ty[:, i]), i]) for i in range(p)])   # population standard deviation, as lab's B.std
        y_eval = y * std + mean
        out["train_x"] = tx.tolist()
        out["train_y"] = [[None if np.isnan(v) else v for v in row] for row in ty.tolist()]
    out["logpdf"] = gpar_ref.gpar_logpdf(x, y_eval, w, hypers, c["config"], impute=False, replace=False, eps=eps)
    return out


def workload_case(name, c, rng):
    n, p = c["n"], c["p"]
    x = np.sort(rng.uniform(0.0, 1.0, n))
    base = np.stack([np.sin(5 * x + k) + 0.3 * k * x for k in range(p)], axis=1)
    y = base + 0.1 * rng.standard_normal((n, p))
    if c["pattern"] == "scattered":
        y[rng.random((n, p)) < 0.2] = np.nan
        y[0] = base[0]
    elif c["pattern"] == "tail-block":
        y[n - 8 :, p - 2 :] = np.nan  # the last two outputs are unobserved on the last rows (eeg.py holds out F1 / F2 / FZ)
    else:
        y[8:14, 0] = np.nan
        y[12:20, 2] = np.nan
    hypers = hypers_for(1, p, c["config"], rng)
    x_ind = None if c["x_ind"] is None else np.linspace(0.0, 1.0, c["x_ind"])
    value = gpar_ref.gpar_logpdf(x, y, None, hypers, c["config"], impute=c["impute"], replace=c["replace"], eps=c["epsilon"], x_ind=x_ind)
    return {"name": name, "config": c["config"], "impute": c["impute"], "replace": c["replace"], "epsilon": c["epsilon"],
            "x_ind": None if x_ind is None else x_ind.tolist(), "x": x[:, None].tolist(),
            "y": [[None if np.isnan(v) else v for v in row] for row in y.tolist()], "w": None, "hypers": hypers, "logpdf": value}


def main():
    rng = np.random.default_rng(20260928)
    out = {"gpar_logpdf": [], "single_gp": [], "vfe": []}
    for name, c in CASES:
        n, m, p = c["n"], c["m"], c["p"]
        x = rng.uniform(-1.5, 1.5, (n, m))
        y = rng.standard_normal((n, p))
        if c["missing"]:
            y[rng.random((n, p)) < c["missing"]] = np.nan
            y[0] = rng.standard_normal(p)  # at least one complete row
        w = (rng.random((n, p)) + 0.5) if c["weights"] else None
        hypers = hypers_for(m, p, c["config"], rng)
        value = gpar_ref.gpar_logpdf(x, y, w, hypers, c["config"], impute=c["impute"], replace=c["replace"])
        out["gpar_logpdf"].append(
            {
                "name": name, "config": c["config"], "impute": c["impute"], "replace": c["replace"],
                "x": x.tolist(), "y": [[None if np.isnan(v) else v for v in row] for row in y.tolist()],
                "w": None if w is None else w.tolist(), "hypers": hypers, "logpdf": value,
            }
        )
    # single-layer posterior moments for each kernel family
    for name, config in [("eq", dict(linear=False)), ("rq", dict(linear=False, rq=True)),
                         ("per", dict(linear=False, per=True)), ("inlin", dict(linear=False, input_linear=True))]:
        n, m = 14, 2
        x, xs = rng.uniform(-1, 1, (n, m)), rng.uniform(-1, 1, (6, m))
        y = rng.standard_normal(n)
        noise = rng.uniform(0.05, 0.2, n)
        hypers = hypers_for(m, 1, config, rng)
        spec, _ = gpar_ref.layer_spec(hypers, m, 0, config)
        mean, cov = gp_ref.posterior(spec, x, y, noise, xs)
        out["single_gp"].append({"name": name, "config": config, "hypers": hypers, "x": x.tolist(), "y": y.tolist(),
                                 "noise": noise.tolist(), "xs": xs.tolist(), "logpdf": gp_ref.logpdf(spec, x, y, noise),
                                 "mean": mean.tolist(), "cov": cov.tolist()})
    # inducing points
    for name, nz in [("vfe-few", 5), ("vfe-many", 12)]:
        n, m = 30, 1
        config = dict(linear=False)
        x, xs, z = rng.uniform(-2, 2, (n, m)), rng.uniform(-2, 2, (7, m)), np.linspace(-2, 2, nz)[:, None]
        y = np.sin(2 * x[:, 0]) + 0.1 * rng.standard_normal(n)
        noise = rng.uniform(0.05, 0.1, n)
        hypers = hypers_for(m, 1, config, rng)
        spec, _ = gpar_ref.layer_spec(hypers, m, 0, config)
        mean, cov = gp_ref.vfe_posterior(spec, x, y, noise, z, xs)
        out["vfe"].append({"name": name, "config": config, "hypers": hypers, "x": x.tolist(), "y": y.tolist(), "noise": noise.tolist(),
                           "z": z.tolist(), "xs": xs.tolist(), "bound": gp_ref.vfe_bound(spec, x, y, noise, z),
                           "mean": mean.tolist(), "cov": cov.tolist()})
    # appended after everything else so that the earlier vectors keep their values (one shared random stream)
    wl_rng = np.random.default_rng(20260929)
    for name, c in WORKLOADS:
        out["gpar_logpdf"].append(workload_case(name, c, wl_rng))
    micro_rng = np.random.default_rng(20260930)   # (a stream of its own: the vectors above keep their values)
    for name, c in MICRO:
        out["gpar_logpdf"].append(micro_case(name, c, micro_rng))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpar_cases.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
