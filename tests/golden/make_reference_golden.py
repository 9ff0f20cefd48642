"""Generates tests/golden/reference_cases.json: the cases of gpar_cases.json evaluated by the REAL reference implementation.

This script imports `gpar` (wesselb/gpar, e.g. from /root/reference) together with its dependencies stheno / lab / matrix /
varz / plum / wbml.  Those are not installable in the offline build container (no package index), so `reference_cases.json` is
absent from this repository and the oracle is "parity unpinned" (oracle/__init__.py).  Wherever the
dependencies ARE available, one command pins it:

    pip install stheno varz backends backends-matrix plum-dispatch wbml      # plus torch, numpy
    PYTHONPATH=/path/to/wesselb-gpar python tests/golden/make_reference_golden.py

and `tests/test_reference_golden.py` then checks the CPU oracle (and, with `-m gpu`, the HIP path) against the reference's
own numbers.  Nothing of the reference is copied: the script only IMPORTS it, feeds it the inputs and hyper-parameters the
committed cases already hold, and writes numbers.

For every case of gpar_cases.json (same `name`, same x / y / w / hypers / config):
    gpar_logpdf[]   GPARRegressor(**config).logpdf(x, y, w)            /root/reference/gpar/regression.py:461-506
    single_gp[]     f(x, noise).logpdf(y); (f | (f(x, noise), y)) mean and covariance at xs    gpar/model.py:226,298-301
    vfe[]           PseudoObs(f(z), f(x, noise), y): elbo, posterior mean / covariance at xs   gpar/model.py:286-287
The `micro-*` cases among gpar_logpdf[] each isolate ONE detail of the third-party arithmetic that the build recalled rather than
read (order of the periodic embedding's scales, RQ at alpha = 1e-2, the additive constant of the input-linear term, noise / w,
where the jitter goes, the Markov window, the un-normalising logpdf quirk with the population standard deviation): a mismatch in
reference_cases.json names the detail (tests/golden/make_golden.py: MICRO).
plus, for every gpar_logpdf case, the posterior mean of `predict`-style conditioning is NOT recorded (it is sampled there);
the deterministic pieces above are what the reference's own tests pin by identities (tests/test_regression.py:92-137).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _set_variables(reg, hypers):
    """Pre-create the regressor's variables with the case's values (varz get-or-create: the constructor's inits are then
    ignored), with the bounds the reference's model generator uses (gpar/regression.py:92-180)."""
    for name, value in hypers.items():
        value = np.asarray(value, dtype=np.float64)
        if name.endswith("/input/lin/const"):
            reg.vs.get(value, name=name)
        elif name.endswith("/alpha"):
            reg.vs.bnd(value, lower=1e-3, upper=1e3, name=name)
        elif name.endswith("/noise"):
            reg.vs.bnd(value, lower=1e-8, name=name)
        else:
            reg.vs.bnd(value, name=name)


def _nan(rows):
    return np.array([[np.nan if v is None else v for v in row] for row in rows], dtype=np.float64)


def main():
    import torch
    from lab.torch import B
    from stheno import Obs, PseudoObs

    from gpar.regression import GPARRegressor, _construct_gpar

    with open(os.path.join(HERE, "gpar_cases.json")) as f:
        cases = json.load(f)
    out = {"generator": "wesselb/gpar + stheno (see header of make_reference_golden.py)", "versions": {}, "gpar_logpdf": [], "single_gp": [], "vfe": []}
    for mod in ("gpar", "stheno", "lab", "matrix", "varz", "mlkernels", "torch", "numpy"):
        try:
            out["versions"][mod] = getattr(__import__(mod), "__version__", "?")
        except Exception:  # noqa: BLE001
            out["versions"][mod] = None

    def layer0(config, hypers, m):
        reg = GPARRegressor(normalise_y=False, **config)
        _set_variables(reg, hypers)
        return _construct_gpar(reg, reg.vs, m, 1).layers[0]()

    for case in cases["gpar_logpdf"]:
        B.epsilon = case.get("epsilon", 1e-12)
        x_ind = case.get("x_ind")
        reg = GPARRegressor(replace=case["replace"], impute=case["impute"], normalise_y="train_y" in case,
                            x_ind=None if x_ind is None else np.array(x_ind), **case["config"])
        _set_variables(reg, case["hypers"])
        if "train_y" in case:   # micro-normalise-quirk: conditioned (normalise_y=True) before the PRIOR logpdf is asked for
            reg.condition(np.array(case["train_x"]), _nan(case["train_y"]))
        w = None if case["w"] is None else np.array(case["w"])
        value = reg.logpdf(np.array(case["x"]), _nan(case["y"]), w)
        out["gpar_logpdf"].append({"name": case["name"], "logpdf": float(value)})
    B.epsilon = 1e-12
    to_t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))
    for case in cases["single_gp"]:
        f, _ = layer0(case["config"], case["hypers"], 2)
        x, y, noise, xs = (to_t(case[k]) for k in ("x", "y", "noise", "xs"))
        post = f | Obs(f(x, noise), y[:, None])
        out["single_gp"].append({"name": case["name"], "logpdf": float(f(x, noise).logpdf(y[:, None])),
                                 "mean": B.to_numpy(B.dense(post.mean(xs)))[:, 0].tolist(),
                                 "cov": B.to_numpy(B.dense(post.kernel(xs))).tolist()})
    for case in cases["vfe"]:
        f, _ = layer0(case["config"], case["hypers"], 1)
        x, y, noise, z, xs = (to_t(case[k]) for k in ("x", "y", "noise", "z", "xs"))
        obs = PseudoObs(f(z), f(x, noise), y[:, None])
        post = f | obs
        out["vfe"].append({"name": case["name"], "bound": float(obs.elbo(f.measure)),
                           "mean": B.to_numpy(B.dense(post.mean(xs)))[:, 0].tolist(),
                           "cov": B.to_numpy(B.dense(post.kernel(xs))).tolist()})
    path = os.path.join(HERE, "reference_cases.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    try:
        import stheno  # noqa: F401
        import gpar  # noqa: F401
    except ImportError as exc:
        sys.exit(f"make_reference_golden.py needs the reference and its dependencies (see the header): {exc}")
    main()
