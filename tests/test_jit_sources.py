"""The run-time specialised kernels are generated source (csrc/gram_jit.h): `gpar_jit_compile_check` compiles - without loading,
so without a GPU - the kernel of a given layer structure for gfx950.  Every kernel family GPARRegressor can build must compile."""
import ctypes
import os

import numpy as np
import pytest


def _specs():
    from gpar_amd.engine import set_engine
    from gpar_amd.kernels import compile_kernel
    from gpar_amd.regression import GPARRegressor, _construct_gpar
    from oracle.engine import OracleEngine

    previous = set_engine(OracleEngine())
    try:
        out = []
        for kw, m, p in [
            (dict(scale=0.5, linear=True, nonlinear=True, markov=2), 4, 8),                       # BASELINE C3
            (dict(scale=0.5, linear=True), 2, 4),                                                 # C2
            (dict(scale=0.5, per=True, rq=True, linear=True, nonlinear=True), 3, 16),             # C5 (42 feature dims in the last layer)
            (dict(rq=True, input_linear=True, linear=True, nonlinear=True, scale_tie=True), 2, 3),
            (dict(linear=True, nonlinear=True, markov=0), 1, 3),                                  # constant output term
        ]:
            reg = GPARRegressor(**kw)
            gpar = _construct_gpar(reg, reg.vs, m, p)
            for layer in sorted({0, 1, p - 1}):
                f, _ = gpar.layers[layer]()
                out.append(compile_kernel(f.kernel, m + layer))
        return out
    finally:
        set_engine(previous)


def test_every_layer_structure_compiles_for_gfx950():
    """Kinds (include/gpar_hip.h): 0 Gram build - narrow structures the strip kernel, wide ones (C5's 42 dims) the 4 x 4 micro-tile form
    with rolled dim loops (round 5), more than 48 dims refused -, 1 / 21 parameter-gradient pass (21: with the frequency
    derivatives of periodic features), 2 input-gradient pass."""
    from gpar_amd import _lib
    from gpar_amd.engine import GRAM_JIT_MAX_DZ, GRAM_JIT_WIDE_MAX_DZ

    lib = _lib.load()
    log = ctypes.create_string_buffer(1 << 16)
    sizes, wide = [], 0
    for ck in _specs():
        size = lib.gpar_jit_compile_check(0, ctypes.byref(ck.kspec), ck.dz, b"gfx950", log, len(log))
        if ck.dz > GRAM_JIT_WIDE_MAX_DZ:
            assert size < -1000   # refused: the interpreter serves it
        else:
            assert size > 0, log.value.decode()[:4000]
            sizes.append(size)
            wide += ck.dz > GRAM_JIT_MAX_DZ
        periodic = any(f.periods is not None for t in ck.kernel.terms for f in t.factors)
        grad = lib.gpar_jit_compile_check(21 if periodic else 1, ctypes.byref(ck.kspec), ck.dz, b"gfx950", log, len(log))
        assert grad > 0, log.value.decode()[:4000]
        if 1 <= ck.dz <= 20:
            assert lib.gpar_jit_compile_check(2, ctypes.byref(ck.kspec), ck.dz, b"gfx950", log, len(log)) > 0, log.value.decode()[:4000]
    assert len(sizes) >= 10 and len(set(sizes)) > 3 and wide >= 1   # different structures, different code


def test_bad_arguments_are_refused():
    from gpar_amd import _lib

    lib = _lib.load()
    ck = _specs()[0]
    log = ctypes.create_string_buffer(4096)
    assert lib.gpar_jit_compile_check(99, ctypes.byref(ck.kspec), ck.dz, b"gfx950", log, len(log)) < -1000   # unknown kind
    assert lib.gpar_jit_compile_check(0, ctypes.byref(ck.kspec), 1000, b"gfx950", log, len(log)) < -1000     # more dims than the spec holds
    assert lib.gpar_jit_compile_check(0, None, ck.dz, b"gfx950", log, len(log)) < -1000


def test_build_time_archive_covers_the_baseline_configurations():
    """gpar_amd/aot.py: the archive next to the library holds the Gram / gradient kernels of the common layer structures - among
    them every layer of BASELINE C2, C3 and C4 and the gradient passes of C5 - under the keys the library derives at run time."""
    from gpar_amd import aot

    assert os.path.exists(aot.ARCHIVE), "build it: python -c 'import __graft_entry__ as g; g.build()'"
    arch, keys = aot.read_keys()
    assert arch == "gfx950" and len(keys) == len(set(keys)) >= 300
    kinds = {k.split("#")[0] for k in keys}
    assert kinds == {"0", "1", "2"}   # Gram, parameter-gradient, input-gradient
    wanted = {(kind, bytes(ck.kspec), ck.dz) for kind, ck in aot.jobs()}
    assert len(wanted) == len(keys)
    # a structure that no keyword family produces is not there: the library compiles it at run time, as before
    assert not any("d90x" in k for k in keys)
