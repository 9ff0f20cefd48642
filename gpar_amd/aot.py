"""Build step: the archive of layer kernels compiled at BUILD time (csrc/jit.h: gpar_aot_<arch>.bin next to libgpar_hip.so).

A GPAR layer's kernel STRUCTURE - which terms, factor types, feature offsets and dim counts - follows from the regressor's
keywords, the number of inputs m and the layer index (reference gpar/regression.py:92-180); the generated Gram / gradient kernels
depend on nothing else.  The structures of the common keyword combinations are enumerated here, each is compiled with the very
call the library makes at run time (hiprtc, same options; needs no GPU) and the code objects are written into one archive that
the library consults before it compiles anything.  Structures that are not in it are compiled at run time as before.

    python -m gpar_amd.aot [--jobs N]        (__graft_entry__.build() runs this when the archive is older than the library)
"""
import ctypes
import os
import struct
import sys
import time

ARCH = "gfx950"
ARCHIVE = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"gpar_aot_{ARCH}.bin")
MAGIC = b"GPARAOT2"   # + u32 ABI version + u64 generator fingerprint (csrc/jit.h): the library ignores an archive that is not its own

# keyword families (everything else at its default); the BASELINE configurations are members: C2 = "linear" at m = 2, C3 = "nonlinear"
# with markov = 2 at m = 4, C4 = "nonlinear" at m = 8 (+ the sparse kinds), C5 = "per_rq" at m = 3
FAMILIES = {
    "linear": dict(linear=True, nonlinear=False),
    "nonlinear": dict(linear=True, nonlinear=True),
    "nonlinear_only": dict(linear=False, nonlinear=True),
    "rq": dict(linear=True, nonlinear=True, rq=True),
    "per": dict(linear=True, nonlinear=True, per=True),
    "per_rq": dict(linear=True, nonlinear=True, per=True, rq=True),
}


def structures():
    """{structure: (compiled kernel, periodic?, sparse kinds wanted?)} over the families, m = 1 .. 4 and the first eight layers (with
    and without a Markov order), plus the deeper / wider layers of the BASELINE configurations."""
    import torch

    from .kernels import compile_kernel
    from .regression import GPARRegressor, _construct_gpar

    cases = []
    for name, kw in FAMILIES.items():
        for m in (1, 2, 3, 4):
            cases.append((kw, None, m, 8, name in ("linear", "nonlinear") and m <= 2))
            cases.append((kw, 2, m, 4, False))
    cases.append((FAMILIES["nonlinear"], 2, 4, 8, False))     # C3
    cases.append((FAMILIES["nonlinear"], None, 8, 4, True))   # C4 (inducing points: rectangular weights, input gradients)
    cases.append((FAMILIES["per_rq"], None, 3, 16, False))    # C5
    out = {}
    with torch.no_grad():
        for kw, markov, m, p, sparse in cases:
            reg = GPARRegressor(markov=markov, **kw)
            gpar = _construct_gpar(reg, reg.vs, m, p)
            for i, model in enumerate(gpar.layers):
                f, _ = model()
                ck = compile_kernel(f.kernel, m + i)
                ks = ck.kspec
                key = (ck.dz, int(ks.nterms)) + tuple((int(a.type), int(a.term), int(a.off), int(a.nd)) for a in ks.factor[: int(ks.nfactors)])
                periodic = any(fa.periods is not None for t in ck.kernel.terms for fa in t.factors)
                if key in out:
                    out[key] = (out[key][0], periodic, out[key][2] or sparse)
                else:
                    out[key] = (ck, periodic, sparse)
    return out


def jobs():
    """(kind, compiled kernel) for every kernel the archive holds: the Gram build (narrow structures) and the parameter-gradient pass
    for all; rectangular-weight and input-gradient passes where inducing points / joint training are common."""
    from .engine import GRAM_JIT_WIDE_MAX_DZ

    todo = []
    for ck, periodic, sparse in structures().values():
        zd = 20 if periodic else 0
        if ck.dz <= GRAM_JIT_WIDE_MAX_DZ:   # (narrow: the strip kernel; wider than GRAM_JIT_MAX_DZ: the 4 x 4 micro-tile form)
            todo.append((0, ck))
        todo.append((1 + zd, ck))
        if sparse:
            todo.append((11 + zd, ck))
            if 1 <= ck.dz <= 20:
                todo += [(2, ck), (12, ck)]
    return todo


def _compile(job):
    from . import _lib

    kind, ks_bytes, dz = job
    lib = _lib.load()
    ks = _lib.KSpec.from_buffer_copy(ks_bytes)
    cap = 1 << 20
    code = ctypes.create_string_buffer(cap)
    key = ctypes.create_string_buffer(1024)
    log = ctypes.create_string_buffer(1 << 14)
    size = lib.gpar_jit_compile(kind, ctypes.byref(ks), dz, ARCH.encode(), code, cap, key, len(key), log, len(log))
    if size <= 0 or size > cap:
        return None, f"kind {kind} dz {dz}: {size} {log.value.decode(errors='replace')[:300]}"
    return (key.value, code.raw[:size]), None


def build(jobs_n=None, quiet=False):
    """Compile every kernel of `jobs()` (one process per core: hiprtc serialises compilations inside a process) and write the archive."""
    from concurrent.futures import ProcessPoolExecutor

    t0 = time.time()
    todo = [(kind, bytes(ck.kspec), ck.dz) for kind, ck in jobs()]
    workers = jobs_n or min(len(todo), max(1, (os.cpu_count() or 2)))
    entries, errors = {}, []
    with ProcessPoolExecutor(max_workers=workers) as pool:
        for got, err in pool.map(_compile, todo, chunksize=4):
            if err:
                errors.append(err)
            else:
                entries[got[0]] = got[1]
    if errors:
        raise RuntimeError(f"{len(errors)} kernels failed to compile, e.g. {errors[0]}")
    from . import _lib

    lib = _lib.load()
    tmp = ARCHIVE + ".tmp"
    with open(tmp, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<IQ", lib.gpar_abi_version(), lib.gpar_aot_fingerprint()))
        f.write(struct.pack("<I", len(ARCH)) + ARCH.encode())
        f.write(struct.pack("<I", len(entries)))
        for key in sorted(entries):
            f.write(struct.pack("<I", len(key)) + key)
            f.write(struct.pack("<Q", len(entries[key])) + entries[key])
    os.replace(tmp, ARCHIVE)
    if not quiet:
        print(f"[aot] {len(entries)} kernels, {os.path.getsize(ARCHIVE) / 1e6:.1f} MB, {time.time() - t0:.0f} s with {workers} processes -> {ARCHIVE}", flush=True)
    return len(entries)


def read_header(path=ARCHIVE):
    """(ABI version, generator fingerprint) an archive was built for, or None if there is no readable archive of this format."""
    try:
        with open(path, "rb") as f:
            head = f.read(20)
    except OSError:
        return None
    if len(head) < 20 or head[:8] != MAGIC:
        return None
    return struct.unpack_from("<IQ", head, 8)


def is_current(path=ARCHIVE):
    """Does the archive belong to the library next to it?  (The library applies the same test before it uses one.)"""
    from . import _lib

    lib = _lib.load()
    return read_header(path) == (lib.gpar_abi_version(), lib.gpar_aot_fingerprint())


def read_keys(path=ARCHIVE):
    """Keys of an archive (for tests)."""
    with open(path, "rb") as f:
        blob = f.read()
    assert blob[:8] == MAGIC
    at = 8 + 12
    (alen,) = struct.unpack_from("<I", blob, at)
    at += 4
    arch = blob[at:at + alen].decode()
    at += alen
    (count,) = struct.unpack_from("<I", blob, at)
    at += 4
    keys = []
    for _ in range(count):
        (klen,) = struct.unpack_from("<I", blob, at)
        at += 4
        keys.append(blob[at:at + klen].decode())
        at += klen
        (size,) = struct.unpack_from("<Q", blob, at)
        at += 8 + size
    return arch, keys


if __name__ == "__main__":
    n = None
    if "--jobs" in sys.argv:
        n = int(sys.argv[sys.argv.index("--jobs") + 1])
    build(n)
