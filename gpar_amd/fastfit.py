"""The training objective of ONE dense layer with fixed inputs, off the host's critical path.

`GPARRegressor.fit(fix=True)` (reference gpar/regression.py:418-459) minimises, layer by layer, the negative log marginal
likelihood of a GP whose inputs no longer depend on anything being optimised.  Per L-BFGS-B function call the general route
(`optimise.minimise_l_bfgs_b` over `GPAR.logpdf`) rebuilds the layer's kernel objects from the variable store, walks the model,
allocates the factor's buffers and lets torch autograd chain the analytic device gradient through the bound transforms -
0.7-1.0 ms of host time per evaluation, against 0.3-0.9 ms for a whole four-layer `logpdf` at n <= 2048.

Here the layer is prepared ONCE: its kernel expression is built over mutable value holders (one per hyper-parameter of the store),
its device buffers (features, augmented matrix, inverse, weights, partial sums, results) are allocated once, and an evaluation is

    latent vector -> bounded values (the store's own transform) written into the holders -> `compile_kernel` (the ctypes
    specification the library takes by value) -> ONE library call (`gpar_logpdf_dense_grad`: Gram, augmented Cholesky, value,
    inverse, weights, fused weighted-sum pass) -> ONE device-to-host copy of [value, log det, moment sums, 1/2 diag W, info] ->
    the chain rule on the host in numpy (`HipEngine._grads_from_moments`, then the bounded transforms' derivatives).

Same device launches on the same inputs as the general route (`gp.Obs._value` with `_fuse_grad`): the value is the same to the
bit, the gradient to rounding of the host-side chain rule.  No autograd, no model objects, no allocation per evaluation.  Anything
this route does not cover (inducing points, trainable inducing inputs, `fix=False`, a posterior process, more rows than
`one_call_grad_rows()`, a retry ladder, an engine without the library) returns None from `build` and the caller keeps the general
route; a failed factorisation (non-positive pivot, expired hand-off spin) is handed to the general route's evaluation, which owns
the unfused retry (`model._retry_unfused`).
"""
import ctypes
import logging
import os
import threading

import numpy as np
import scipy.optimize
import torch

from . import _lib, hip, optimise
from .kernels import Kernel, compile_kernel

__all__ = ["build", "DenseLayerObjective", "LockstepFactor", "lockstep_rows"]

log = logging.getLogger(__name__)


class _Holder(np.ndarray):
    """The current (bounded) value of one hyper-parameter of the store, as a mutable numpy array the kernel expression holds.
    Only the object created here carries the name; anything derived from it (a product, a view) does not."""

    def __new__(cls, value, name):
        obj = np.array(value, dtype=np.float64, copy=True).view(cls)
        obj.gpar_name = name
        return obj

    def __array_finalize__(self, obj):
        self.gpar_name = None

    # `holder * kernel` must reach Kernel.__rmul__ with the holder itself (numpy would unwrap a 0-d array to a float first)
    def __mul__(self, other):
        return NotImplemented if isinstance(other, Kernel) else np.ndarray.__mul__(self, other)

    def __add__(self, other):
        return NotImplemented if isinstance(other, Kernel) else np.ndarray.__add__(self, other)


class _TracingVars:
    """What `regression._model_generator` needs of a variable store (`bnd`, `get`, `memo`), handing out holders instead of tensors.
    Variables are created in the real store on first use with the reference's initialisations and bounds, exactly as the general
    route's first evaluation would."""

    def __init__(self, vs):
        self.vs = vs
        self.holders = {}

    def _holder(self, name):
        h = self.holders.get(name)
        if h is None:
            h = self.holders[name] = _Holder(self.vs[name].detach().numpy(), name)
        return h

    def bnd(self, init=None, lower=1e-4, upper=1e4, name=None, shape=None):
        with torch.no_grad():
            self.vs.bnd(init=init, lower=lower, upper=upper, name=name, shape=shape)
        return self._holder(name)

    bounded = bnd

    def get(self, init=None, name=None, shape=None):
        with torch.no_grad():
            self.vs.get(init=init, name=name, shape=shape)
        return self._holder(name)

    unbounded = get

    def pos(self, init=None, name=None, shape=None):
        with torch.no_grad():
            self.vs.pos(init=init, name=name, shape=shape)
        return self._holder(name)

    positive = pos

    def memo(self, key, build_fn):
        return build_fn()


def _slots(kernel):
    """[(kind, term index, factor index, holder or float)] of every parameter position of the kernel; None if a position holds
    something whose origin is unknown (a product of two variables: not produced by the reference's layer kernels)."""
    out = []
    for ti, term in enumerate(kernel.terms):
        entries = [("coef", ti, None, term.coef)]
        for fi, f in enumerate(term.factors):
            entries += [(kind, ti, fi, getattr(f, kind)) for kind in ("scales", "periods", "alpha")]
        for kind, ti_, fi, v in entries:
            if v is None or isinstance(v, (int, float)):
                continue
            if isinstance(v, _Holder) and v.gpar_name is not None:
                out.append((kind, ti_, fi, v))
            else:
                return None
    return out


def lockstep_rows():
    """(min, max) rows between which concurrently trained layers factor their matrices TOGETHER (LockstepFactor).  OFF by default:
    (1, 0).  Measured on three boxes of the pool (profiles/r06_lockstep_fit.txt; fit(iters=20), four layers, rendezvous off -> on):
    n = 1024 37 -> 38 / 37 -> 48 ms, 1536 52 -> 68 / 50 -> 56, 2048 61 -> 71 / 60 -> 70 / 60 -> 60, 2560 87 -> 89 / 88 -> 108, 3072 121 -> 119 /
    111 -> 99 / 113 -> 128, 4096 220 -> 210 / 208 -> 199 / 210 -> 229.  A round costs every lane the slowest lane's evaluation plus two
    thread hand-offs, whose latency is the host's; free-running lanes hide one lane's factorisation under the others' inverses.  Where the
    factorisation dominates (from ~3000 rows) the rendezvous wins 5-11 % on two boxes and loses 9-14 % on the third: not a default.
    GPAR_FIT_LOCKSTEP_ROWS=lo[:hi] switches it on between lo and hi rows (hi at most `gp.one_call_grad_rows()`, above which the
    prepared objective does not apply; 0 = never)."""
    from .gp import one_call_grad_rows

    env = os.environ.get("GPAR_FIT_LOCKSTEP_ROWS")
    if env is None:
        return (1, 0)
    parts = env.split(":")
    lo, hi = int(parts[0]), one_call_grad_rows()
    if lo <= 0:
        return (1, 0)
    if len(parts) > 1:
        hi = min(hi, int(parts[1]))
    return lo, hi


class LockstepFactor:
    """The lock-step training rendezvous: the layers `fit(fix=True)` trains independently of one another (reference
    gpar/regression.py:418-446 runs their optimisations one after the other) evaluate their objectives in ROUNDS, and the
    factorisations of a round are ONE gpar_potrf_batch.

    Why: from ~1000 rows on a factorisation is one launch of hundreds of workgroups that wait for one another, and such launches
    of different streams must not be in flight together (csrc/panel.h: the spin chain) - four training threads queue their
    factorisations one behind the other (n = 4096: 4 x 1.2 ms per round of evaluations), while the same four matrices factored
    in lock-step take 2.7 ms.  Everything else of an evaluation (Gram build before, inverse / weights / gradient pass after) stays
    on the member's own stream and overlaps with the other members'.  What it costs: the lanes now move in step - a round lasts as
    long as its slowest evaluation plus two thread hand-offs - where free-running lanes hide one lane's factorisation under the
    others' inverses; whether it pays depends on the host (`lockstep_rows`: off by default).

    Protocol (deterministic: the composition of a round depends on the members' evaluation counts, never on timing).  Every lane
    (host thread + stream) owns one slot of a shared (lanes (n + 1)) x (n + 1) buffer.  Per evaluation a lane enqueues its build into
    its slot, records an event and ARRIVES; when every active lane has arrived, the last one makes its stream wait for the others'
    events, enqueues gpar_potrf_batch over each run of consecutive arrived slots, records an event behind it and wakes the others,
    which make their streams wait for that event and go on with gpar_logpdf_dense_grad_finish.  A lane LEAVES when its last layer is
    trained (or when it falls back to the general route for good); leaving may complete a round the others were waiting for.
    The factor of a matrix inside a batch of k equals the factor of the matrix alone to rounding, not to the bit (csrc/potrf.h: the
    schedule is a function of the shape AND the batch), so a concurrently trained model equals the serially trained one to the
    optimiser's amplification of that (1e-8 .. 1e-6 relative after 20 iterations), not bit for bit."""

    def __init__(self, eng, n, lanes):
        self.n, self.lanes = int(n), int(lanes)
        self.cond = threading.Condition()
        self.active = set(range(self.lanes))
        self.arrived = {}          # lane -> event behind its build
        self.generation = 0
        self.done = None           # event behind the batch of the generation that completed last
        self.error = None
        self.rounds = 0
        self.batches = 0
        self.history = []          # (lanes of the round), for tests: the composition of every round
        self._allocate(eng)

    # ---- the device side (replaced by the CPU tests of the protocol: tests/test_fastfit.py) -----------------------------------
    def _allocate(self, eng):
        dev = eng.device
        self.lib = _lib.load()
        self.A = hip.alloc_matrix(self.lanes * (self.n + 1), self.n + 1, dev)
        self.lda = hip._ld(self.A)
        self.stride = (self.n + 1) * self.lda            # elements between consecutive slots
        self.logdet = torch.zeros(self.lanes, dtype=torch.float64, device=dev)
        self.info = torch.zeros(self.lanes, dtype=torch.int32, device=dev)

    def _stream(self):
        return torch.cuda.current_stream(self.A.device)

    def _record(self, stream):
        event = torch.cuda.Event()
        event.record(stream)
        return event

    def _wait(self, stream, event):
        stream.wait_event(event)

    def _potrf(self, stream, first, count):
        a, ld, info = self.slot(first)
        rc = self.lib.gpar_potrf_batch(a, count, self.stride, self.n + 1, self.n, self.lda, ld, info, 0, stream.cuda_stream)
        _lib.check(rc, "gpar_potrf_batch")

    # pointers of a lane's slot
    def slot(self, lane):
        return (self.A.data_ptr() + 8 * lane * self.stride, self.logdet.data_ptr() + 8 * lane, self.info.data_ptr() + 4 * lane)

    # ---- the protocol --------------------------------------------------------------------------------------------------------
    def _launch(self):
        """(lock held, every active lane has arrived) one gpar_potrf_batch per run of consecutive slots, on the caller's stream."""
        lanes = sorted(self.arrived)
        try:
            stream = self._stream()
            for lane in lanes:
                self._wait(stream, self.arrived[lane])
            i = 0
            while i < len(lanes):
                j = i
                while j + 1 < len(lanes) and lanes[j + 1] == lanes[j] + 1:
                    j += 1
                self._potrf(stream, lanes[i], j - i + 1)
                self.batches += 1
                i = j + 1
            self.done = self._record(stream)
        except BaseException as exc:  # noqa: BLE001 - every waiting lane must hear of it
            self.error = exc
        self.history.append(tuple(lanes))
        self.arrived = {}
        self.generation += 1
        self.rounds += 1
        self.cond.notify_all()

    def factor(self, lane):
        """The lane's build is enqueued on the current stream: factor its slot with the round; on return the current stream is
        ordered behind the factorisation."""
        stream = self._stream()
        mine = self._record(stream)
        with self.cond:
            if lane not in self.active:
                raise RuntimeError("lane is not a member of the rendezvous")
            self.arrived[lane] = mine
            gen = self.generation
            if set(self.arrived) >= self.active:
                self._launch()
            else:
                while self.generation == gen:
                    self.cond.wait()
            if self.error is not None:
                raise self.error
            done = self.done
        self._wait(stream, done)

    def leave(self, lane):
        """The lane takes no further part (idempotent)."""
        with self.cond:
            if lane not in self.active:
                return
            self.active.discard(lane)
            self.arrived.pop(lane, None)
            if self.active and set(self.arrived) >= self.active:
                self._launch()


class DenseLayerObjective:
    """-log N(y; 0, K_theta(X) + noise / w + eps I) of layer `pi` and its gradient with respect to the latent (unconstrained)
    variables `names` of `vs`, evaluated as described in the module docstring.  X (n x width), y, w (n) are device tensors that do
    not change during the optimisation."""

    def __init__(self, eng, vs, names, kernel, noise_holder, holders, X, y, w, general_fg=None, group=None, lane=None):
        self.eng, self.vs, self.names = eng, vs, list(names)
        self.group, self.lane = group, lane   # a LockstepFactor and this objective's slot in it, or None
        self.kernel, self.noise, self.holders = kernel, noise_holder, holders
        self.general_fg = general_fg
        self.X = X if X.stride(-1) == 1 else X.contiguous()
        self.n, self.width = int(self.X.shape[0]), int(self.X.shape[1])
        self.y = y.reshape(-1).contiguous()
        self.w = w.reshape(-1).contiguous()
        w_host = self.w.cpu().numpy()
        self.unit_weights = bool(np.all(w_host == 1.0))
        self.inv_w = None if self.unit_weights else 1.0 / w_host
        # the variables being optimised: (name, kind, lower, upper, slice of the latent vector, shape)
        self.layout, at = [], 0
        for name in self.names:
            var = vs._vars[name]
            size = int(var.latent.numel())
            self.layout.append((name, var.kind, var.lower, var.upper, slice(at, at + size), tuple(var.latent.shape)))
            at += size
        self.size = at
        self.slots = _slots(kernel)
        self.evaluations = 0
        self.fallbacks = 0
        self._allocate(compile_kernel(kernel, self.width))

    # ---- the device side (overridden by the CPU tests of the host logic: tests/test_fastfit.py) ----------------------------------
    def _allocate(self, ck):
        """Every device buffer of an evaluation, once; the pointers of the library call, once."""
        eng, n, dev = self.eng, self.n, self.X.device
        self.lib = _lib.load()
        hip._check_mat(self.X, "X")
        self.periodic = eng._periodic(ck)
        dz = max(ck.dz, 1)
        self.z = hip.alloc_matrix(n, dz, dev)
        self.zd = hip.alloc_matrix(n, dz, dev, zero=True) if self.periodic else None
        if self.group is not None and self.group.n != n:
            raise ValueError("the rendezvous was made for another number of rows")
        self.A = hip.alloc_matrix(n + 1, n + 1, dev) if self.group is None else None   # (a member's factor lives in its slot of the group's buffer)
        self.Xw = hip.alloc_matrix(n, n, dev)
        self.W = hip.alloc_matrix(n, n, dev)
        nt = (n + 63) // 64
        self.nblocks = max(1, min(nt * (nt + 1) // 2, 1024))
        nacc = _lib.GRAD_NACC
        self.work = torch.empty(self.nblocks * nacc + n, dtype=torch.float64, device=dev)
        # [value, log det, moment sums (nacc), 1/2 diag W (n), info word (int32 in the last 8 bytes)]
        self.res = torch.zeros(2 + nacc + n + 1, dtype=torch.float64, device=dev)
        self.res_host = torch.zeros(2 + nacc + n + 1, dtype=torch.float64).pin_memory()
        self.res_np = self.res_host.numpy()
        self.info_np = self.res_np[-1:].view(np.int32)
        self.noise_vec = torch.empty(n, dtype=torch.float64, device=dev)
        self.noise_num = torch.empty((), dtype=torch.float64, device=dev)
        self.flags = 0
        self._ptrs = dict(
            x=self.X.data_ptr(), ldx=hip._ld(self.X), y=self.y.data_ptr(), incy=int(self.y.stride(0)), noise=self.noise_vec.data_ptr(),
            z=self.z.data_ptr(), zd=None if self.zd is None else self.zd.data_ptr(), ldz=hip._ld(self.z),
            A=None if self.A is None else self.A.data_ptr(), lda=None if self.A is None else hip._ld(self.A),
            X=self.Xw.data_ptr(), ldxw=hip._ld(self.Xw), W=self.W.data_ptr(), ldw=hip._ld(self.W),
            alpha=self.work[self.nblocks * nacc:].data_ptr(), work=self.work.data_ptr(), out=self.res.data_ptr(),
            half=self.res[2 + nacc:].data_ptr(), info=self.res[2 + nacc + n:].data_ptr(),
        )

    def _device_eval(self, ck, noise):
        """One library call + one device-to-host copy: (log marginal likelihood, kernel-parameter gradients of it, 1/2 diag W as
        a host vector), or None when the factorisation reported a failure."""
        dev = self.X.device
        stream = torch.cuda.current_stream(dev)
        if self.unit_weights:
            self.noise_vec.fill_(noise)
        else:
            # the correctly rounded quotient of two tensors, as model.GPAR._noise_over forms it
            torch.true_divide(self.noise_num.fill_(noise), self.w, out=self.noise_vec)
        p = self._ptrs
        if self.group is None:
            rc = self.lib.gpar_logpdf_dense_grad(
                ctypes.byref(ck.fspec), ctypes.byref(ck.kspec), p["x"], self.n, p["ldx"], p["y"], p["incy"], p["noise"], float(self.eng.epsilon),
                p["z"], p["zd"], p["ldz"], p["A"], p["lda"], p["X"], p["ldxw"], p["W"], p["ldw"], p["alpha"], p["work"], self.nblocks, p["out"],
                p["half"], p["info"], self.flags, stream.cuda_stream)
            _lib.check(rc, "gpar_logpdf_dense_grad")
        else:
            # build into this lane's slot, factor with the round, finish on this stream (LockstepFactor)
            g = self.group
            a, ld, info = g.slot(self.lane)
            rc = self.lib.gpar_logpdf_dense_build(
                ctypes.byref(ck.fspec), ctypes.byref(ck.kspec), p["x"], self.n, p["ldx"], p["y"], p["incy"], p["noise"], float(self.eng.epsilon),
                p["z"], p["ldz"], a, g.lda, ld, info, stream.cuda_stream)
            _lib.check(rc, "gpar_logpdf_dense_build")
            g.factor(self.lane)
            rc = self.lib.gpar_logpdf_dense_grad_finish(
                ctypes.byref(ck.fspec), ctypes.byref(ck.kspec), p["x"], self.n, p["ldx"], p["z"], p["zd"], p["ldz"], a, g.lda, ld, info,
                p["X"], p["ldxw"], p["W"], p["ldw"], p["alpha"], p["work"], self.nblocks, p["out"], p["half"], p["info"], stream.cuda_stream)
            _lib.check(rc, "gpar_logpdf_dense_grad_finish")
        self.res_host.copy_(self.res, non_blocking=True)
        stream.synchronize()
        if int(self.info_np[0]) != 0:
            return None
        nacc = _lib.GRAD_NACC
        res = self.res_np
        return float(res[0]), self.eng._grads_from_moments(ck, res[2:2 + nacc], 0.5), res[2 + nacc:2 + nacc + self.n]

    # ---- one evaluation -----------------------------------------------------------------------------------------------------
    def _write_values(self, x):
        """Latent vector -> bounded values into the holders (the store's transform: lower + (upper - lower) sigmoid(latent)); returns
        the sigmoids for the chain rule.  The sigmoid is torch's, taken PER VARIABLE on a tensor of the variable's own shape - exactly the
        call `vars._Var.value` makes: one call over the whole vector would send some elements down torch's vectorised path and others
        down its scalar tail, whose exponentials can differ in the last bit, and the general route's values would no longer be
        reproduced bit for bit (found by tools/r06/fuzz_fastfit.py: 1 evaluation in ~200 differed by 8e-13).  One torch call per
        variable, on views of two persistent buffers."""
        views = getattr(self, "_views", None)
        if views is None:
            self._latent_np = np.zeros(self.size)
            self._sig_np = np.zeros(self.size)
            latent, sig = torch.from_numpy(self._latent_np), torch.from_numpy(self._sig_np)
            views = self._views = [(latent[sl].reshape(shape), sig[sl].reshape(shape)) if kind in ("bnd", "pos") else None
                                   for _, kind, _, _, sl, shape in self.layout]
        self._latent_np[:] = x
        sig = self._sig_np
        for (name, kind, lower, upper, sl, shape), view in zip(self.layout, views):
            if kind == "bnd":
                torch.sigmoid(view[0], out=view[1])
            elif kind == "pos":
                torch.exp(view[0], out=view[1])   # (for "pos" the buffer holds exp(latent): value and derivative at once)
            h = self.holders.get(name)
            if h is None:
                continue   # (a selected variable the layer does not read: its gradient is zero)
            if kind == "bnd":
                h[...] = (lower + (upper - lower) * sig[sl]).reshape(shape)
            elif kind == "pos":
                h[...] = sig[sl].reshape(shape)
            else:
                h[...] = self._latent_np[sl].reshape(shape)
        return sig

    def fg(self, x):
        """(objective, gradient) at the latent vector x: what scipy's L-BFGS-B calls."""
        optimise.count_evaluation()
        self.evaluations += 1
        x = np.asarray(x, dtype=np.float64)
        sig = self._write_values(x)
        ck = compile_kernel(self.kernel, self.width)
        got = self._device_eval(ck, float(self.noise))
        if got is None:
            # a non-positive pivot or an expired hand-off spin: the general route owns the confirmation on the unfused path
            self.fallbacks += 1
            if self.general_fg is not None:
                return self.general_fg(x)
            return np.nan, np.zeros_like(x)
        value, grads, half = got
        by_name = {}

        def add(holder, val):
            val = np.asarray(val, dtype=np.float64)
            if val.size != holder.size:
                val = val.sum()   # a scalar parameter broadcast over several features
            prev = by_name.get(holder.gpar_name)
            val = np.asarray(val, dtype=np.float64).reshape(holder.shape)
            by_name[holder.gpar_name] = val if prev is None else prev + val

        for kind, ti, fi, holder in self.slots:
            add(holder, grads["coef"][ti] if kind == "coef" else grads["factors"][ti][fi][kind])
        if isinstance(self.noise, _Holder) and self.noise.gpar_name is not None:
            add(self.noise, half.sum() if self.inv_w is None else float(np.dot(half, self.inv_w)))
        grad = np.zeros(self.size)
        for name, kind, lower, upper, sl, shape in self.layout:
            g = by_name.get(name)
            if g is None:
                continue
            g = g.reshape(-1)
            if kind == "bnd":
                s = sig[sl]
                g = ((g * (upper - lower)) * (1.0 - s)) * s   # (the order autograd multiplies in: same bits as the general route)
            elif kind == "pos":
                g = g * sig[sl]
            grad[sl] = -g
        return -value, grad

    # ---- the optimisation ----------------------------------------------------------------------------------------------------
    def minimise(self, iters=1000, f_calls=10000, trace=False):
        """L-BFGS-B over the selected latents (as optimise.minimise_l_bfgs_b); the optimum is written back into the store."""
        x0 = self.vs.get_vector(self.names)

        def fg(x):
            val, grad = self.fg(x)
            if trace:
                print(f"  objective {val:.6e}  |grad| {np.linalg.norm(grad):.3e}")
            return val, grad

        x_opt, val, _ = scipy.optimize.fmin_l_bfgs_b(fg, x0, maxiter=iters, maxfun=f_calls)
        self.vs.set_vector(x_opt, self.names)
        return val


def build(reg, eng, vs, pi, names, fixed_x, item, general_fg=None, cls=None, group=None, lane=None):
    """The fast objective of layer `pi` of regressor `reg` with fixed design matrix `fixed_x`, or None when this route does not
    apply.  `item` = (y_i, w_i, mask) as `model.per_output` yields it for output pi; `names` the variable names being optimised."""
    from .gp import one_call_grad_rows
    from .regression import _model_generator

    on_device = cls is None   # (the CPU tests of the host logic hand in a class with its own device side)
    cls = DenseLayerObjective if cls is None else cls
    if reg.sparse or getattr(reg, "_x_ind_trainable", False) or (on_device and not hasattr(eng, "_grads_from_moments")):
        return None
    if getattr(eng, "cholesky_retry_factor", 1.0) > 1.0 or getattr(getattr(eng, "_tls", None), "safe", False):
        return None
    if not isinstance(fixed_x, torch.Tensor) or (on_device and not fixed_x.is_cuda) or fixed_x.requires_grad:
        return None
    yi, wi, mask = item
    if not isinstance(yi, torch.Tensor) or (on_device and not yi.is_cuda):
        return None
    X = fixed_x[mask]
    keep = getattr(yi, "_obs_idx", None)
    n_missing = getattr(yi, "_n_missing", None)
    if not isinstance(mask, slice):
        if n_missing is None:
            return None   # (the pattern is not planned on the host: the general route finds the rows)
        if n_missing:
            X, yi, wi = X.index_select(0, keep), yi.index_select(0, keep), wi.index_select(0, keep)
    n = int(X.shape[0])
    if not 0 < n <= one_call_grad_rows():
        return None
    tracer = _TracingVars(vs)
    f, noise = _model_generator(tracer, reg.m, pi, **reg.model_config)()
    kernel = f.kernel
    if not isinstance(kernel, Kernel) or f.is_posterior or _slots(kernel) is None:
        return None
    if not isinstance(noise, _Holder):
        return None
    names = vs.match(names)   # (globs; resolved now that the layer's variables exist)
    if not names:
        return None
    if group is not None and (group.n != n or not on_device):
        group = None
    return cls(eng, vs, names, kernel, noise, tracer.holders, X, yi, wi.reshape(-1), general_fg=general_fg, group=group,
               lane=lane if group is not None else None)
