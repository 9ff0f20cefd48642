import os, sys
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/tests') else os.getcwd())
import numpy as np, torch
from tests.test_hip_primitives import _jit_cases
from gpar_amd import hip as H
from gpar_amd.kernels import compile_kernel
from oracle import kernels as ok
dev = torch.device("cuda:0")
for case in range(5):
    name, kernel, width = _jit_cases()[case]
    ck = compile_kernel(kernel, width)
    g = torch.Generator().manual_seed(case)
    x = torch.randn(333, width, generator=g, dtype=torch.float64).to(dev)
    z = H.featurize(ck, x)
    out = {}
    for mode in ["-1", "0"]:
        os.environ["GPAR_GRAM_JIT_MIN_ENTRIES"] = mode
        out[mode] = torch.tril(H.gram(ck, z, None, lower=True, diag_const=0.1)).cpu().numpy()
    d = np.abs(out["0"] - out["-1"])
    want = np.tril(ok.gram(ok.spec_to_dict(kernel.resolve(width)), x.cpu().numpy(), None, jitter=0.1))
    i = np.unravel_index(np.argmax(d), d.shape)
    print(name, "dz", ck.dz, "max |jit - interp|", d.max(), "at", i, out["0"][i], out["-1"][i], " vs oracle: jit", np.abs(out["0"] - want).max(), "interp", np.abs(out["-1"] - want).max(), "n differing", int((d > 0).sum()))
