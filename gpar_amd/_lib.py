"""ctypes binding of libgpar_hip.so (C ABI: include/gpar_hip.h).

There is deliberately no fallback: if the shared library is missing or does not export the ABI the header
declares, importing the binding raises.  The library lives in-tree next to this file (built by
`__graft_entry__.build()` / `python -m gpar_amd.build`).
"""
import ctypes
import os

GPAR_MAX_DIMS = 96
GPAR_MAX_FACTORS = 12
GPAR_MAX_TERMS = 8

EMBED_ID, EMBED_SIN, EMBED_COS = 0, 1, 2
K_EQ, K_RQ, K_LINEAR = 0, 1, 2

GRAM_LOWER = 1
GEMM_C_LOWER = 1
GEMM_A_LOWER = 2
GEMM_K_FROM_ROW = 4
GEMM_K_TO_COL = 8
POTRF_NO_LOOKAHEAD = 1
POTRF_UNFUSED = 2
WS_GEMM_SPLITK, WS_GEMV_T, WS_GRAM_GRAD, WS_CHOL_INVERSE, WS_INPUT_GRAD = 1, 2, 3, 4, 5
GRAD_NACC = GPAR_MAX_TERMS + GPAR_MAX_FACTORS + 2 * GPAR_MAX_DIMS

ABI_VERSION = 7

LIB_NAME = "libgpar_hip.so"
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)


class FSpec(ctypes.Structure):
    _fields_ = [
        ("dz", ctypes.c_int32),
        ("pad_", ctypes.c_int32),
        ("col", ctypes.c_int32 * GPAR_MAX_DIMS),
        ("embed", ctypes.c_int32 * GPAR_MAX_DIMS),
        ("inv_scale", ctypes.c_double * GPAR_MAX_DIMS),
        ("freq", ctypes.c_double * GPAR_MAX_DIMS),
    ]


class Factor(ctypes.Structure):
    _fields_ = [
        ("type", ctypes.c_int32),
        ("term", ctypes.c_int32),
        ("off", ctypes.c_int32),
        ("nd", ctypes.c_int32),
        ("alpha", ctypes.c_double),
    ]


class KSpec(ctypes.Structure):
    _fields_ = [
        ("nterms", ctypes.c_int32),
        ("nfactors", ctypes.c_int32),
        ("coef", ctypes.c_double * GPAR_MAX_TERMS),
        ("factor", Factor * GPAR_MAX_FACTORS),
    ]


class Layer(ctypes.Structure):
    """gpar_layer_t: one layer of a lock-step evaluation (pointers to host-side specifications)."""
    _fields_ = [
        ("fs", ctypes.POINTER(FSpec)),
        ("ks", ctypes.POINTER(KSpec)),
        ("noise", ctypes.c_double),
        ("y_col", ctypes.c_int32),
        ("pad_", ctypes.c_int32),
    ]


_c_int = ctypes.c_int
_c_dbl = ctypes.c_double
_ptr = ctypes.c_void_p
_u64 = ctypes.c_uint64

# name -> (restype, argtypes); every symbol include/gpar_hip.h declares
SIGNATURES = {
    "gpar_abi_version": (_c_int, []),
    "gpar_sizeof_fspec": (ctypes.c_size_t, []),
    "gpar_sizeof_kspec": (ctypes.c_size_t, []),
    "gpar_jit_compile_check": (_c_int, [_c_int, ctypes.POINTER(KSpec), _c_int, ctypes.c_char_p, ctypes.c_char_p, _c_int]),
    "gpar_init": (_c_int, [_ptr]),
    "gpar_jit_prepare": (_c_int, [_c_int, ctypes.POINTER(KSpec), _c_int, _ptr]),
    "gpar_jit_compile": (ctypes.c_longlong, [_c_int, ctypes.POINTER(KSpec), _c_int, ctypes.c_char_p, _ptr, ctypes.c_longlong, ctypes.c_char_p, _c_int,
                                             ctypes.c_char_p, _c_int]),
    "gpar_aot_stats": (_c_int, [ctypes.POINTER(_c_int), ctypes.POINTER(_c_int)]),
    "gpar_aot_fingerprint": (ctypes.c_ulonglong, []),
    "gpar_jit_stats": (_c_int, [ctypes.POINTER(_c_int), ctypes.POINTER(_c_int), ctypes.POINTER(_c_int)]),
    "gpar_featurize": (_c_int, [ctypes.POINTER(FSpec), _ptr, _c_int, _c_int, _ptr, _c_int, _ptr]),
    "gpar_gram": (
        _c_int,
        [ctypes.POINTER(KSpec), _ptr, _c_int, _c_int, _ptr, _c_int, _c_int, _c_int, _ptr, _c_int, _c_int, _ptr, _c_dbl, _ptr, _ptr],
    ),
    "gpar_gram_batch": (
        _c_int,
        [ctypes.POINTER(KSpec), _ptr, _c_int, _c_int, ctypes.c_longlong, _c_int, _ptr, _c_int, ctypes.c_longlong, _c_int, _ptr, _c_dbl, _c_int,
         _ptr],
    ),
    "gpar_gram_diag": (_c_int, [ctypes.POINTER(KSpec), _ptr, _c_int, _c_int, _c_int, _ptr, _ptr]),
    "gpar_featurize_dfreq": (_c_int, [ctypes.POINTER(FSpec), _ptr, _c_int, _c_int, _ptr, _c_int, _ptr]),
    "gpar_grad_nacc": (_c_int, []),
    "gpar_gram_grad": (
        _c_int,
        [ctypes.POINTER(KSpec), _ptr, _ptr, _c_int, _c_int, _c_int, _ptr, _c_int, _ptr, _c_int, _ptr, _ptr],
    ),
    "gpar_gram_grad_cross": (
        _c_int,
        [ctypes.POINTER(KSpec), _ptr, _ptr, _c_int, _c_int, _ptr, _ptr, _c_int, _c_int, _c_int, _ptr, _c_int, _c_int, _ptr, _c_int,
         _ptr, _ptr],
    ),
    "gpar_gram_input_grad": (
        _c_int,
        [ctypes.POINTER(KSpec), _ptr, _c_int, _c_int, _ptr, _c_int, _c_int, _c_int, _ptr, _c_int, _c_int, _c_int, _ptr, _ptr, _c_int, _ptr],
    ),
    "gpar_logpdf_dense": (
        _c_int,
        [ctypes.POINTER(FSpec), ctypes.POINTER(KSpec), _ptr, _c_int, _c_int, _ptr, ctypes.c_long, _ptr, _c_dbl, _ptr, _c_int, _ptr, _c_int,
         _ptr, _ptr, _ptr, _c_int, _ptr],
    ),
    "gpar_logpdf_dense_grad": (
        _c_int,
        [ctypes.POINTER(FSpec), ctypes.POINTER(KSpec), _ptr, _c_int, _c_int, _ptr, ctypes.c_long, _ptr, _c_dbl, _ptr, _ptr, _c_int, _ptr, _c_int,
         _ptr, _c_int, _ptr, _c_int, _ptr, _ptr, _c_int, _ptr, _ptr, _ptr, _c_int, _ptr],
    ),
    "gpar_logpdf_dense_grad_finish": (
        _c_int,
        [ctypes.POINTER(FSpec), ctypes.POINTER(KSpec), _ptr, _c_int, _c_int, _ptr, _ptr, _c_int, _ptr, _c_int, _ptr, _ptr, _ptr, _c_int, _ptr, _c_int,
         _ptr, _ptr, _c_int, _ptr, _ptr, _ptr, _ptr],
    ),
    "gpar_logpdf_dense_build": (
        _c_int,
        [ctypes.POINTER(FSpec), ctypes.POINTER(KSpec), _ptr, _c_int, _c_int, _ptr, ctypes.c_long, _ptr, _c_dbl, _ptr, _c_int, _ptr, _c_int,
         _ptr, _ptr, _ptr],
    ),
    "gpar_sizeof_layer": (ctypes.c_size_t, []),
    "gpar_logpdf_lockstep": (
        _c_int,
        [ctypes.POINTER(Layer), _c_int, _ptr, _c_int, _c_int, _ptr, _c_int, _ptr, _c_int, _c_dbl, _ptr, _c_int, _ptr, _ptr, _c_int,
         ctypes.c_longlong, _ptr, _ptr, _ptr, _ptr, _c_int, _ptr],
    ),
    "gpar_potrf_batch": (_c_int, [_ptr, _c_int, ctypes.c_longlong, _c_int, _c_int, _c_int, _ptr, _ptr, _c_int, _ptr]),
    "gpar_logpdf_dense_finish": (_c_int, [_ptr, _c_int, ctypes.c_longlong, _c_int, _c_int, _ptr, _ptr, _ptr]),
    "gpar_potrf": (_c_int, [_ptr, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr]),
    "gpar_potrf_ex": (_c_int, [_ptr, _c_int, _c_int, _c_int, _ptr, _ptr, _c_int, _ptr]),
    "gpar_trsm_rlt": (_c_int, [_ptr, _c_int, _c_int, _ptr, _c_int, _c_int, _ptr]),
    "gpar_trsm_rlt_if": (_c_int, [_ptr, _c_int, _c_int, _ptr, _c_int, _c_int, _ptr, _c_int, _ptr]),
    "gpar_vfe_assemble": (_c_int, [_ptr, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _c_int, _c_dbl, _ptr, _c_int, _ptr, _ptr, _ptr, _ptr]),
    "gpar_vfe_value": (_c_int, [_ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr]),
    "gpar_chol_spread": (_c_int, [_ptr, _c_int, _c_int, _c_dbl, _ptr, _ptr, _ptr]),
    "gpar_trsm_rln": (_c_int, [_ptr, _c_int, _c_int, _ptr, _c_int, _c_int, _ptr]),
    "gpar_chol_inverse": (_c_int, [_ptr, _c_int, _c_int, _ptr, _c_int, _ptr, _c_int, _ptr]),
    "gpar_gemm": (
        _c_int,
        [_c_int, _c_int, _c_int, _c_int, _c_int, _c_dbl, _ptr, _c_int, _ptr, _c_int, _c_dbl, _ptr, _c_int, _c_int, _ptr],
    ),
    "gpar_gemm_batch": (
        _c_int,
        [_c_int, _c_int, _c_int, _c_int, _c_int, _c_dbl, _ptr, _c_int, ctypes.c_longlong, _ptr, _c_int, ctypes.c_longlong, _c_dbl, _ptr,
         _c_int, ctypes.c_longlong, _c_int, _c_int, _ptr],
    ),
    "gpar_gemm_splitk": (
        _c_int,
        [_c_int, _c_int, _c_int, _c_int, _c_int, _c_dbl, _ptr, _c_int, _ptr, _c_int, _c_dbl, _ptr, _c_int, _c_int, _c_int, _ptr, _ptr],
    ),
    "gpar_dot": (_c_int, [_ptr, _c_int, _ptr, _c_int, _c_int, _ptr, _c_int, _ptr]),
    "gpar_gemv_t": (_c_int, [_ptr, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr]),
    "gpar_rownorm2": (_c_int, [_ptr, _c_int, _c_int, _c_int, _ptr, _ptr]),
    "gpar_pack_lower": (_c_int, [_ptr, _c_int, _c_int, _ptr, _ptr]),
    "gpar_unpack_lower": (_c_int, [_ptr, _c_int, _ptr, _c_int, _ptr]),
    "gpar_workspace_doubles": (ctypes.c_longlong, [_c_int, _c_int, _c_int, _c_int]),
    "gpar_randn": (_c_int, [_u64, _u64, _ptr, _c_int, _c_int, _c_int, _ptr]),
    "gpar_trmv_lower": (_c_int, [_ptr, _c_int, _c_int, _ptr, _c_int, _ptr, _c_int, _ptr]),
    "gpar_trmv_upper": (_c_int, [_ptr, _c_int, _c_int, _ptr, _c_int, _ptr, _c_int, _ptr]),
    "gpar_gemv": (_c_int, [_ptr, _c_int, _c_int, _c_int, _ptr, _c_int, _c_dbl, _ptr, _c_int, _ptr]),
    "gpar_trmv_lower_batch": (
        _c_int,
        [_ptr, _c_int, ctypes.c_longlong, _c_int, _c_int, _ptr, _c_int, ctypes.c_longlong, _ptr, _c_int, ctypes.c_longlong, _ptr, _c_int,
         ctypes.c_longlong, _ptr],
    ),
    "gpar_sample_stats": (
        _c_int,
        [_ptr, _c_int, ctypes.c_longlong, ctypes.c_longlong, _c_int, _c_dbl, _c_int, _c_dbl, _ptr, _ptr, _ptr, _ptr],
    ),
    "gpar_profile_enable": (_c_int, [_c_int]),
    "gpar_profile_read": (_c_int, [ctypes.POINTER(_c_int), ctypes.POINTER(_c_dbl), ctypes.POINTER(_c_dbl), ctypes.POINTER(_c_dbl), _c_int]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """Load libgpar_hip.so (once) and bind every declared symbol.  Raises HipLibraryError loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  gpar_amd has no CPU fallback."
        )
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the machine
        raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.gpar_abi_version() != ABI_VERSION:
        raise HipLibraryError(f"ABI version mismatch: library {lib.gpar_abi_version()}, binding {ABI_VERSION}")
    if lib.gpar_sizeof_fspec() != ctypes.sizeof(FSpec) or lib.gpar_sizeof_kspec() != ctypes.sizeof(KSpec):
        raise HipLibraryError("struct layout mismatch between include/gpar_hip.h and gpar_amd/_lib.py")
    if lib.gpar_sizeof_layer() != ctypes.sizeof(Layer):
        raise HipLibraryError("gpar_layer_t layout mismatch between include/gpar_hip.h and gpar_amd/_lib.py")
    if lib.gpar_grad_nacc() != GRAD_NACC:
        raise HipLibraryError(f"gradient accumulator layout mismatch: library {lib.gpar_grad_nacc()}, binding {GRAD_NACC}")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise HipLibraryError(f"{what} failed with status {rc}")
