"""CPU-side checks of the drop-in boundary: the C-ABI shared library loads, exports every symbol that
include/gpar_hip.h declares, agrees with the ctypes binding on struct layouts and the ABI version — and the product
refuses to run without a GPU instead of falling back to anything."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gpar_hip.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|size_t|long long)\s+(gpar_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = _declared_functions()
    for required in ["gpar_gram", "gpar_potrf", "gpar_trsm_rlt", "gpar_trsm_rln", "gpar_gemm", "gpar_featurize",
                     "gpar_gram_grad", "gpar_randn", "gpar_abi_version"]:
        assert required in names


def test_library_exports_every_declared_symbol():
    from gpar_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared_functions():
        assert hasattr(lib, name), f"{name} is declared in include/gpar_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in gpar_amd/_lib.py"
    assert set(_lib.SIGNATURES) == set(_declared_functions())


def test_struct_layouts_and_version_agree():
    from gpar_amd import _lib

    lib = _lib.load()
    assert lib.gpar_abi_version() == _lib.ABI_VERSION
    assert lib.gpar_sizeof_fspec() == ctypes.sizeof(_lib.FSpec)
    assert lib.gpar_sizeof_kspec() == ctypes.sizeof(_lib.KSpec)
    assert lib.gpar_grad_nacc() == _lib.GRAD_NACC
    text = open(HEADER).read()
    for macro, value in [("GPAR_MAX_DIMS", _lib.GPAR_MAX_DIMS), ("GPAR_MAX_FACTORS", _lib.GPAR_MAX_FACTORS), ("GPAR_MAX_TERMS", _lib.GPAR_MAX_TERMS)]:
        assert re.search(rf"#define\s+{macro}\s+{value}\b", text)


def test_integration_document_asserts_the_current_abi_version():
    """INTEGRATION.md shows a maintainer of the reference the ctypes binding, including the version check: the number asserted
    there is GPAR_ABI_VERSION of the header (it once stayed at 4 while the library had moved to 5)."""
    text = open(HEADER).read()
    version = int(re.search(r"#define\s+GPAR_ABI_VERSION\s+(\d+)", text).group(1))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    asserted = [int(v) for v in re.findall(r"gpar_abi_version\(\)\s*==\s*(\d+)", doc)]
    assert asserted and all(v == version for v in asserted), (asserted, version)
    from gpar_amd import _lib

    assert _lib.ABI_VERSION == version


def test_build_time_kernel_archive_is_tied_to_the_library():
    """The archive of build-time compiled kernels carries the ABI version and a fingerprint of the library's kernel generators
    (csrc/jit.h); the library ignores an archive whose header is not its own - file dates decide nothing."""
    import struct

    from gpar_amd import _lib, aot

    lib = _lib.load()
    fingerprint = lib.gpar_aot_fingerprint()
    assert fingerprint != 0 and fingerprint == lib.gpar_aot_fingerprint()
    if os.path.exists(aot.ARCHIVE):
        assert aot.read_header() == (lib.gpar_abi_version(), fingerprint)
        assert aot.is_current()
    # a header with another fingerprint is not current
    import tempfile

    with tempfile.NamedTemporaryFile(suffix=".bin") as f:
        f.write(aot.MAGIC + struct.pack("<IQ", lib.gpar_abi_version(), fingerprint ^ 1) + struct.pack("<I", 6) + b"gfx950" + struct.pack("<I", 0))
        f.flush()
        assert not aot.is_current(f.name)
        assert aot.read_header(f.name) == (lib.gpar_abi_version(), fingerprint ^ 1)


def test_no_cpu_fallback_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from gpar_amd.engine import HipEngine, get_engine, set_engine

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HipEngine()
    previous = set_engine(None)
    try:
        with pytest.raises(RuntimeError):
            get_engine()
    finally:
        set_engine(previous)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gpar_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip")):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn


def test_kernel_compile_limits_and_errors():
    import numpy as np

    from gpar_amd.kernels import EQ, Linear, compile_kernel

    k = (2.0 * EQ().stretch(np.ones(3))).select([0, 1, 2]) + Linear().stretch(np.ones(2)).select([3, 4]) + 0.5
    ck = compile_kernel(k, 5)
    assert (ck.dz, ck.kspec.nterms, ck.kspec.nfactors) == (5, 3, 2)
    assert [ck.fspec.col[i] for i in range(5)] == [0, 1, 2, 3, 4]
    with pytest.raises(ValueError):
        compile_kernel(k, 4)  # selects a column that does not exist
    with pytest.raises(ValueError):
        compile_kernel(EQ().stretch(np.ones(2)).select([0, 1, 2]), 3)  # scales / columns mismatch
    with pytest.raises(NotImplementedError):
        EQ().periodic(1.0).stretch(1.0)
    big = EQ().stretch(np.ones(97)).select(list(range(97)))
    with pytest.raises(ValueError):
        compile_kernel(big, 97)


def test_percentile_index_reproduces_numpy_linear_method():
    """Host half of gpar_sample_stats: (k, g) reproduces np.percentile's default method for every S and q the
    regressor uses (and some it does not) - bit for bit on numpy >= 2.0, to 1e-14 in any case."""
    import numpy as np

    from gpar_amd.hip import percentile_index

    def lerp(a, b, t):  # numpy's _lerp
        d = b - a
        return b - d * (1 - t) if t >= 0.5 else a + d * t

    rng = np.random.default_rng(0)
    for S in [1, 2, 3, 10, 40, 100, 101, 1000]:
        v = np.sort(rng.standard_normal(S))
        for q in [0.0, 2.5, 33.3, 50.0, 97.5, 100.0]:
            k, g = percentile_index(S, q)
            assert 0 <= k <= S - 1
            mine = lerp(v[k], v[min(k + 1, S - 1)], g)
            np.testing.assert_allclose(mine, np.percentile(v, q), rtol=1e-14, atol=1e-300)
            if int(np.__version__.split(".")[0]) >= 2:
                assert mine == np.percentile(v, q)
