"""Kernel algebra for GPAR layers, compiled to the flat spec the HIP Gram kernel consumes.

Host-side counterpart of the mlkernels subset the reference composes per layer
(/root/reference/gpar/regression.py:92-180): `EQ`, `RQ`, `Linear`, `ZeroKernel`, `.stretch`, `.periodic`,
`.select`, `+`, `*` and scalar offsets.  Instead of an expression tree evaluated term by term (one n x n
temporary per node), a kernel here is kept in the normal form

    k(x, y) = sum_t coef_t * prod_{f in t} phi_f(z_f(x), z_f(y)),   z_f = stretch(periodic(select(x)))

which `compile_kernel` lowers to `gpar_fspec_t` (feature map) + `gpar_kspec_t` (sum of products) from
include/gpar_hip.h, so that the whole layer kernel and its noise diagonal are produced by ONE fused device
pass.  Hyper-parameters may be Python floats, numpy arrays or torch tensors (the latter keep their autograd
graph; only their values are lowered).
"""
import math

import numpy as np

from . import _lib

__all__ = ["Kernel", "EQ", "RQ", "Linear", "ZeroKernel", "OneKernel", "compile_kernel", "CompiledKernel", "linear_tail"]


def _value(v):
    """Detached float64 numpy value of a float / ndarray / torch tensor."""
    if v is None:
        return None
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v, dtype=np.float64)


class Factor:
    """One elementary kernel applied to stretched (and possibly periodically embedded) selected columns."""

    __slots__ = ("type", "cols", "scales", "periods", "alpha")

    def __init__(self, type, cols=None, scales=None, periods=None, alpha=None):
        self.type = type
        self.cols = cols  # None: every column of the input
        self.scales = scales  # None: unit scales
        self.periods = periods  # None: no periodic embedding
        self.alpha = alpha

    def copy(self, **changes):
        out = Factor(self.type, self.cols, self.scales, self.periods, self.alpha)
        for k, v in changes.items():
            setattr(out, k, v)
        return out

    # value accessors (also used, duck-typed, by the test oracle's converter)
    def scales_value(self):
        nfeat = self.num_features()
        if self.scales is None:
            return np.ones(nfeat)
        s = _value(self.scales).reshape(-1)
        if s.size == 1 and nfeat != 1:
            s = np.full(nfeat, float(s[0]))
        return s

    def periods_value(self):
        p = _value(self.periods).reshape(-1)
        if p.size == 1 and len(self.cols) != 1:
            p = np.full(len(self.cols), float(p[0]))
        return p

    def alpha_value(self):
        return float(_value(self.alpha))

    def num_features(self):
        if self.cols is None:
            raise ValueError("kernel columns are unresolved; call .select(...) or evaluate through a GP")
        return len(self.cols) * (2 if self.periods is not None else 1)


class Term:
    __slots__ = ("coef", "factors")

    def __init__(self, coef, factors):
        self.coef = coef
        self.factors = list(factors)

    def coef_value(self):
        return float(_value(self.coef))


def _scalar_like(x):
    return isinstance(x, (int, float, np.floating, np.integer)) or (hasattr(x, "shape") and tuple(x.shape) in ((), (1,)))


def _times(a, b):
    """a * b for coefficients, handing a tensor through untouched when the other side is the float 1 (the elementary kernels'
    own coefficient): `var * EQ()` then holds `var` ITSELF, so an in-place update of `var` is seen by the next compilation.
    (A product of two tensors is a new tensor, evaluated when the kernel is built - as in any torch expression.)"""
    if isinstance(a, float) and a == 1.0:
        return b
    if isinstance(b, float) and b == 1.0:
        return a
    return a * b


class Kernel:
    """A kernel in sum-of-products normal form."""

    def __init__(self, terms=()):
        self.terms = list(terms)

    # ---- algebra -------------------------------------------------------------------------------
    def __add__(self, other):
        if isinstance(other, Kernel):
            return Kernel(self.terms + other.terms)
        if _scalar_like(other):
            return Kernel(self.terms + [Term(other, [])])
        return NotImplemented

    __radd__ = __add__

    def __mul__(self, other):
        if isinstance(other, Kernel):
            return Kernel(
                [Term(_times(a.coef, b.coef), a.factors + b.factors) for a in self.terms for b in other.terms]
            )
        if _scalar_like(other):
            return Kernel([Term(_times(other, t.coef), t.factors) for t in self.terms])
        return NotImplemented

    __rmul__ = __mul__

    def _map_factors(self, fn):
        return Kernel([Term(t.coef, [fn(f) for f in t.factors]) for t in self.terms])

    def stretch(self, scales):
        """k(x / scales, y / scales)."""

        def fn(f):
            if f.scales is not None or f.periods is not None:
                raise NotImplementedError("stretch must be applied once, before periodic()")
            return f.copy(scales=scales)

        return self._map_factors(fn)

    def periodic(self, periods):
        """k(phi(x), phi(y)) with phi(x) = [sin(2 pi x / T), cos(2 pi x / T)] (features doubled)."""

        def fn(f):
            if f.periods is not None:
                raise NotImplementedError("periodic applied twice")
            return f.copy(periods=periods)

        return self._map_factors(fn)

    def select(self, cols):
        """k(x[:, cols], y[:, cols])."""
        cols = tuple(int(c) for c in cols)

        def fn(f):
            if f.cols is None:
                return f.copy(cols=cols)
            return f.copy(cols=tuple(cols[c] for c in f.cols))

        return self._map_factors(fn)

    def resolve(self, width):
        """Bind unresolved factors to all `width` input columns."""
        return self._map_factors(lambda f: f if f.cols is not None else f.copy(cols=tuple(range(width))))

    @property
    def is_zero(self):
        return len(self.terms) == 0

    def stamp(self):
        """Cheap fingerprint of the hyper-parameter VALUES the kernel currently holds: (identity, in-place version counter)
        of every torch tensor, the bytes of every numpy array.  Floats are immutable and the tree itself never changes, so an
        equal stamp means `compile_kernel` would produce the same device specification."""
        out = []
        for t in self.terms:
            for v in [t.coef] + [x for f in t.factors for x in (f.scales, f.periods, f.alpha)]:
                if hasattr(v, "_version"):
                    out.append((id(v), v._version))
                elif isinstance(v, np.ndarray):
                    out.append(v.tobytes())
        return tuple(out)

    def hyperparameters(self):
        """Every torch-tensor hyper-parameter the kernel holds (for autograd plumbing), in a fixed order."""
        out = []
        for t in self.terms:
            for v in [t.coef] + [x for f in t.factors for x in (f.scales, f.periods, f.alpha)]:
                if hasattr(v, "requires_grad"):
                    out.append(v)
        return out


def EQ():
    return Kernel([Term(1.0, [Factor("eq")])])


def RQ(alpha):
    return Kernel([Term(1.0, [Factor("rq", alpha=alpha)])])


def Linear():
    return Kernel([Term(1.0, [Factor("linear")])])


def ZeroKernel():
    return Kernel([])


def OneKernel():
    return Kernel([Term(1.0, [])])


_TYPE_CODE = {"eq": _lib.K_EQ, "rq": _lib.K_RQ, "linear": _lib.K_LINEAR}


class CompiledKernel:
    """ctypes structs + bookkeeping for one kernel bound to a design-matrix width."""

    __slots__ = ("fspec", "kspec", "dz", "width", "kernel", "layout")

    def __init__(self, fspec, kspec, dz, width, kernel, layout):
        self.fspec, self.kspec, self.dz, self.width, self.kernel, self.layout = fspec, kspec, dz, width, kernel, layout


def linear_tail(ck, shared_cols):
    """Does every factor of the compiled kernel `ck` that reads a design-matrix column >= `shared_cols` stand ALONE in its product
    term and is it linear, with a positive coefficient?  Then
        k(a, b) = k_base(a, b) + <c(a), c(b)>,   c = sqrt(coef) * (the features of those columns),
    where k_base is the kernel with those features set to zero - what GPAR's default output dependence looks like (reference
    gpar/regression.py:141-146: `linear=True, nonlinear=False` adds (y / s)(y' / s)^T over the previous outputs and nothing
    else).  Returns (feature indices, their sqrt(coef) weights) or None."""
    fs, ks = ck.fspec, ck.kspec
    per_term = {}
    for ti, fi, off, nd in ck.layout:
        per_term[ti] = per_term.get(ti, 0) + 1
    idx, weight = [], []
    for ti, fi, off, nd in ck.layout:
        cols = [int(fs.col[q]) for q in range(off, off + nd)]
        if not any(c >= shared_cols for c in cols):
            continue
        f = ck.kernel.terms[ti].factors[fi]
        coef = float(ks.coef[ti])
        if f.type != "linear" or per_term[ti] != 1 or not coef > 0.0:
            return None
        for q, c in zip(range(off, off + nd), cols):
            if c >= shared_cols:
                idx.append(q)
                weight.append(math.sqrt(coef))
    return (idx, weight) if idx else None


def compile_kernel(kernel, width):
    """Lower `kernel` (bound to inputs with `width` columns) to (gpar_fspec_t, gpar_kspec_t).

    `layout` records, per factor, where its features live: (term index, factor index, offset, nd) — the
    gradient code uses it to scatter per-feature derivatives back to hyper-parameters.
    """
    kernel = kernel.resolve(width)
    fs, ks = _lib.FSpec(), _lib.KSpec()
    if len(kernel.terms) > _lib.GPAR_MAX_TERMS:
        raise ValueError(f"kernel has {len(kernel.terms)} terms; the device spec holds {_lib.GPAR_MAX_TERMS}")
    dz, nf, layout = 0, 0, []
    for ti, term in enumerate(kernel.terms):
        ks.coef[ti] = term.coef_value()
        for fi, f in enumerate(term.factors):
            if nf >= _lib.GPAR_MAX_FACTORS:
                raise ValueError("too many kernel factors for the device spec")
            nd = f.num_features()
            if dz + nd > _lib.GPAR_MAX_DIMS:
                raise ValueError("too many kernel feature dimensions for the device spec")
            for c in f.cols:
                if not 0 <= c < width:
                    raise ValueError(f"kernel selects column {c} of an input with {width} columns")
            scales = f.scales_value()
            if scales.size != nd:
                raise ValueError(f"{scales.size} length scales given for {nd} features")
            ncol = len(f.cols)
            if f.periods is not None:
                periods = f.periods_value()
                if periods.size != ncol:
                    raise ValueError(f"{periods.size} periods given for {ncol} columns")
            for q in range(nd):
                j = q % ncol if ncol else 0
                fs.col[dz + q] = f.cols[j]
                fs.inv_scale[dz + q] = 1.0 / scales[q]
                if f.periods is None:
                    fs.embed[dz + q] = _lib.EMBED_ID
                    fs.freq[dz + q] = 0.0
                else:
                    fs.embed[dz + q] = _lib.EMBED_SIN if q < ncol else _lib.EMBED_COS
                    fs.freq[dz + q] = 2.0 * math.pi / periods[j]
            fac = ks.factor[nf]
            fac.type = _TYPE_CODE[f.type]
            fac.term = ti
            fac.off = dz
            fac.nd = nd
            fac.alpha = f.alpha_value() if f.alpha is not None else 0.0
            layout.append((ti, fi, dz, nd))
            dz += nd
            nf += 1
    fs.dz = dz
    ks.nterms = len(kernel.terms)
    ks.nfactors = nf
    return CompiledKernel(fs, ks, dz, width, kernel, layout)
