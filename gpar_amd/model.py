"""GPAR: the autoregressive stack of GP layers (drop-in for /root/reference/gpar/model.py).

Same public surface as the reference module — `GPAR(replace, impute, x_ind)` with `.add_layer`, `.copy`,
`gpar | (x, y, w)`, `.logpdf(...)`, `.sample(...)`, and the helpers `merge`, `construct_model`, `last`,
`per_output` — but the layers are `gpar_amd.gp.GP` objects whose Gram / Cholesky / solve work runs in the HIP
library.  Layer i models output i given the inputs and the previous outputs: its design matrix is
`[x, y_0 .. y_{i-1}]`, where a previous output column holds observed values, or posterior means where
`impute` (missing rows) / `replace` (observed rows) ask for it (reference: model.py:291-322).

Tensors are torch float64 on the engine's device; numpy inputs are accepted and converted.
"""
import threading

import numpy as np
import torch

from .engine import get_engine
from .engine import joining as _joining
from .gp import Obs, PseudoObs, PseudoObsDTC, PseudoObsFITC, Stacked

__all__ = ["GPAR", "merge", "construct_model", "last", "per_output"]


def host_masks():
    """GPAR_HOST_MASKS=0: missing-data masks as boolean device tensors, computed layer by layer (the first implementation: it
    synchronises about ten times per layer); default: planned once on the host (per_output)."""
    import os

    return os.environ.get("GPAR_HOST_MASKS", "1") != "0"


def one_call_enabled():
    """GPAR_ONE_CALL=0: lock-step evaluations through the per-layer build calls (the round-3 route; same bits)."""
    import os

    return os.environ.get("GPAR_ONE_CALL", "1") != "0"


def _device_nan_pattern(y):
    """NaN pattern of a device tensor as a host array: one small device-to-host copy, i.e. a synchronisation.  Nothing is
    remembered here (round 4 kept the last tensor in a process-global: it pinned the tensor - through a view, its whole base
    storage -, was shared by the training threads without a lock, and could not see writes that do not move torch's version
    counter).  A caller that evaluates the SAME outputs again and again attaches the pattern to its own tensor object instead
    (`GPARRegressor.logpdf`: attribute `_gpar_nan` of the tensor the user passed in; `fit`: `_host_nan` of its device copy) and
    `_prep` takes a pattern that arrives attached."""
    return torch.isnan(y).cpu().numpy()


def _attached_nan_pattern(y):
    """The NaN pattern a caller attached to the tensor (`_host_nan`), if the tensor has not been written in place since
    (`_host_nan_version`: the version counter it was taken at; a pattern attached without one is taken as is - test code)."""
    pattern = getattr(y, "_host_nan", None)
    if pattern is None:
        return None
    version = getattr(y, "_host_nan_version", None)
    return pattern if version is None or version == y._version else None


def _is_torch(a):
    return isinstance(a, torch.Tensor)


def _isnan(a):
    return torch.isnan(a) if _is_torch(a) else np.isnan(a)


def _any_along_rows(a):
    return a.any(dim=1) if _is_torch(a) else a.any(axis=1)


def merge(x, updates, to_update):
    """Return `x` with the entries flagged by the boolean vector `to_update` replaced, in order, by `updates`.
    (reference: model.py:14-44; known answers in tests/test_model.py:30-38)"""
    if _is_torch(x):
        out = x.clone()
        mask = to_update if _is_torch(to_update) else torch.as_tensor(np.asarray(to_update), device=x.device)
        out[mask] = updates if _is_torch(updates) else torch.as_tensor(updates, dtype=x.dtype, device=x.device)
        return out
    out = np.array(x, copy=True)
    out[np.asarray(to_update, dtype=bool)] = np.asarray(updates)
    return out


def _merge_rows(x, updates, rows):
    """`merge` with the rows to replace given as an index tensor (no device synchronisation)."""
    out = x.clone()
    out.index_copy_(0, rows, updates.to(dtype=x.dtype))
    return out


def construct_model(f, noise):
    """A layer is a zero-argument callable giving `(latent process, noise variance)` (reference: model.py:47-57)."""

    def model():
        return f, noise

    return model


def last(xs, select=None):
    """Iterate `(is_last, x)` over `xs`, optionally keeping only the positions in `select`; `is_last` refers to the
    position in the full sequence (reference: model.py:60-93; known answers tests/test_model.py:46-52)."""
    wanted = None if select is None else set(select)
    it = iter(xs)
    try:
        current = next(it)
    except StopIteration:
        return
    index = 0
    for upcoming in it:
        if wanted is None or index in wanted:
            yield False, current
        current, index = upcoming, index + 1
    if wanted is None or index in wanted:
        yield True, current


_RETRYING = threading.local()


def notpd_retry_enabled():
    """GPAR_NOTPD_RETRY (default 1): a factorisation that reports a non-positive pivot on the fused panel path is given a second
    opinion on the unfused one before the error reaches the caller."""
    import os

    return os.environ.get("GPAR_NOTPD_RETRY", "1") != "0"


def _retry_unfused(evaluate, layers=(), inputs=(), rewind=False):
    """evaluate(); once more with the engine in safe mode - separate leaf kernels, no look-ahead, no layer pipelining - should

    * a hand-off inside a persistent panel kernel have timed out (device-side info < 0: results invalid).  Whether this retry is
      allowed is read off the actual graph, not off the global grad mode (which is on by default on every inference path): an
      evaluation none of whose layers carries trainable tensors and none of whose inputs requires grad is not part of an
      objective and is simply repeated; an objective under autograd is not retried here - the optimiser treats the error as a
      failed evaluation;
    * a factorisation have reported a non-positive pivot.  K_zz + 1e-12 of many inducing inputs on one axis sits at the edge of
      numerical definiteness: its smallest eigenvalue (~1e-12) is of the size of ANY Cholesky's backward error (2e-15 |K|), and
      which of two backward-stable factorisations gets through is a matter of rounding - LAPACK itself fails on a tenth of such
      matrices (tools/r04_marginal_potrf.py: 6 of 60).  The fused panel kernel solves its strips through explicit inverses of
      16 x 16 blocks with one refinement step (backward error 2.8e-15 where substitution, the unfused path and LAPACK, leave
      1.9e-15) and so fails on a few more: two of forty sparse fuzz cases where numpy's Cholesky does not.  The unfused path
      is the arithmetic of LAPACK's; an error it repeats is reported (for `fit`: a failed evaluation, as in varz).  The
      evaluation builds its graph afresh, so this retry is taken under autograd too."""
    from .engine import HandOffTimeoutError, NotPositiveDefiniteError

    if getattr(_RETRYING, "active", False):
        return evaluate()  # already inside the second attempt of an enclosing evaluation
    calls = getattr(get_engine(), "_calls", None) if rewind else None
    try:
        return evaluate()
    except (HandOffTimeoutError, NotPositiveDefiniteError) as e:
        eng = get_engine()
        if not hasattr(eng, "safe_mode"):
            raise
        if isinstance(e, NotPositiveDefiniteError):
            if not notpd_retry_enabled():
                raise
        elif torch.is_grad_enabled():
            if any(_is_torch(t) and t.requires_grad for t in inputs):
                raise
            if any(_differentiable(*model()) for model in layers):
                raise
        if calls is not None:
            eng._calls = calls  # `rewind`: the repetition draws the same random numbers as the failed attempt
        _RETRYING.active = True
        try:
            with eng.safe_mode():
                return evaluate()
        finally:
            _RETRYING.active = False


def _differentiable(f, noise):
    """Does the layer's log-likelihood take part in an autograd graph (the objective of `fit`)?"""
    from .gp import kernel_parameters

    return (_is_torch(noise) and noise.requires_grad) or bool(kernel_parameters(f.kernel))


def per_output(y, w=None, keep=False):
    """Per layer: `(y_i (n_i x 1), w_i (n_i,), mask_i)` where `mask_i` selects, among the rows that survived
    layer i-1, those observed at output i (or, with `keep`, at any later output: rows needed to keep the data
    closed downwards).  A dict `{keep: [items...]}` is accepted in place of `y` as a precomputed cache.
    (reference: model.py:325-368; known answers tests/test_model.py:55-105)"""
    if isinstance(y, dict):
        yield from y[keep]
        return
    p = y.shape[1]
    host_nan = _attached_nan_pattern(y) if _is_torch(y) else None
    if host_nan is not None and host_nan.shape == tuple(y.shape):
        yield from _per_output_planned(y, w, ~host_nan, keep)
        return
    available = ~_isnan(y)
    if _is_torch(y) and y.is_cuda and bool(available.all()):
        # complete data (one host sync to find out): every mask is "all rows"; a slice instead of a boolean tensor
        # keeps the per-layer loop free of the sync that boolean indexing forces, so the host can enqueue layer
        # i + 1 while the GPU factorises layer i
        for i in range(p):
            yield y[:, i : i + 1], w[:, i], slice(None)
        return
    for i in range(p):
        mask = available[:, i]
        if keep and i < p - 1:
            mask = mask | _any_along_rows(available[:, i + 1 :])
        yield y[mask, i : i + 1], w[mask, i], mask
        y, w, available = y[mask], w[mask], available[mask]


def _per_output_planned(y, w, available, keep):
    """per_output for device tensors whose NaN pattern is known on the host (`available`, numpy bool n x p): the same items, with
    integer index tensors where per_output has boolean masks (same rows, same order; no device synchronisation), a slice where no
    row is dropped and nothing is missing, and - on every y_i - the rows observed / missing at output i as index tensors
    (`_obs_idx`, `_miss_idx`, `_n_missing`) for GPAR._obs and GPAR._update_inputs."""
    p = y.shape[1]
    dev = y.device

    def index(rows):
        return torch.as_tensor(np.ascontiguousarray(rows), dtype=torch.long, device=dev)

    if available.all():
        for i in range(p):
            yi = y[:, i : i + 1]
            yi._n_missing = 0
            yield yi, w[:, i], slice(None)
        return
    for i in range(p):
        mask = available[:, i].copy()
        if keep and i < p - 1:
            mask |= available[:, i + 1 :].any(axis=1)
        rows = np.nonzero(mask)[0]
        all_kept = rows.size == mask.size
        if not all_kept:
            sel = index(rows)
            y, w, available = y.index_select(0, sel), w.index_select(0, sel), available[mask]
        observed = available[:, i]
        n_missing = int(observed.size - observed.sum())
        yi = y[:, i : i + 1]
        yi._n_missing = n_missing
        if n_missing:
            yi._obs_idx = index(np.nonzero(observed)[0])
            yi._miss_idx = index(np.nonzero(~observed)[0])
        if all_kept and n_missing == 0:
            sel = slice(None)
        elif all_kept:
            sel = index(rows)   # (every row stays, but some are missing at this output: not the "complete" case a slice stands for)
        yield yi, w[:, i], sel


def _lockstep_values(eng, pending):
    """Log marginal likelihoods of layers that do not feed one another and share their number of rows: one lock-step batch
    (HipEngine.logpdf_dense_batch) when every one of them qualifies for the value-only path, one after the other otherwise."""
    obs = [o for _, o in pending]
    if len(obs) < 2 or not all(o._value_only() for o in obs) or len({int(o.fdd.n) for o in obs}) != 1:
        return [f.measure.logpdf(o) for f, o in pending]
    n = int(obs[0].fdd.n)
    cap = max(1, eng.batch_bytes() // (8 * (n + 1) * (n + 17)))   # layers per batch within the workspace budget
    values = []
    for i in range(0, len(obs), cap):
        chunk = obs[i:i + cap]
        if len(chunk) < 2:
            values.extend(pending[i + j][0].measure.logpdf(o) for j, o in enumerate(chunk))
            continue
        items = [(eng.compile(o.base.kernel, o.fdd.x.shape[1]), o.fdd.x, o.y, o.fdd.noise) for o in chunk]
        try:
            vals, info = eng.logpdf_dense_batch(items, eng.epsilon)
        except torch.cuda.OutOfMemoryError:   # the batch buffer did not fit beside what the caller holds: one layer at a time
            torch.cuda.empty_cache()
            values.extend(pending[i + j][0].measure.logpdf(o) for j, o in enumerate(chunk))
            continue
        eng.check_info(info)
        values.extend(vals[b].detach() for b in range(len(chunk)))
    return values


def _lockstep_total(eng, x, y, w, entries):
    """Sum of the log marginal likelihoods of layers that do not feed one another and see all rows, by the one-call path
    (HipEngine.logpdf_lockstep), as a list of device scalars to be added in order: `entries` holds (layer index, process, noise) per layer; layer i's design matrix is the first
    m + i columns of [x, y_0 .. y_(p-2)] and its kernel selects among them, so ONE widest matrix serves every layer - no design
    matrix per layer, no per-layer tensor for the noise diagonal, no observation objects.  None when the workspace does not fit
    (the caller falls back to the layer-by-layer route)."""
    m = int(x.shape[1])
    n = int(x.shape[0])
    widest = max(i for i, _, _ in entries)
    x_full = x if widest == 0 else torch.cat([x, y[:, :widest]], dim=1)
    layers = []
    for i, f, noise in entries:
        value = float(noise.detach()) if _is_torch(noise) else float(noise)
        layers.append((eng.compile(f.kernel, m + i), value, i))
    cap = max(1, eng.batch_bytes() // (8 * (n + 1) * (n + 17)))   # layers per batch within the workspace budget
    out = []
    for s0 in range(0, len(layers), cap):
        try:
            values, part, info = eng.logpdf_lockstep(layers[s0:s0 + cap], x_full, y, w, eng.epsilon)
        except torch.cuda.OutOfMemoryError:
            torch.cuda.empty_cache()
            return None
        eng.check_info(info)
        if len(layers) <= cap:
            return [part]   # one batch: the device-side sum in layer order is the sum the caller would form
        out.extend(values[b] for b in range(values.shape[0]))
    return out


def _lockstep_factors(eng, obs):
    """Factor the observations of layers that do not feed one another in lock-step batches (HipEngine.factor_dense_batch) and
    hand each its factor; whatever does not qualify is factored on its own."""
    from .gp import _Factor

    todo = [o for o in obs if o._fac is None]
    if len(todo) >= 2 and all(o._batchable() for o in todo) and len({int(o.fdd.n) for o in todo}) == 1:
        n = int(todo[0].fdd.n)
        cap = max(1, eng.batch_bytes() // (8 * (n + 1) * (n + 17)))
        for i in range(0, len(todo), cap):
            chunk = todo[i:i + cap]
            if len(chunk) < 2:
                break
            items = [(eng.compile(o.base.kernel, o.fdd.x.shape[1]), o.fdd.x, o.y, o.fdd.noise) for o in chunk]
            try:
                A, logdet, info = eng.factor_dense_batch(items, eng.epsilon)
            except torch.cuda.OutOfMemoryError:   # no room for the batch buffer: the loop below factors them one at a time
                torch.cuda.empty_cache()
                break
            eng.check_info(info)
            for b, o in enumerate(chunk):
                o._fac = _Factor.from_batch(eng, n, A[b * (n + 1):(b + 1) * (n + 1)], logdet[b:b + 1])
    for o in obs:
        o.factor()


class GPAR:
    """Gaussian process autoregressive model.

    Args:
        replace (bool): feed posterior means instead of observations to later layers.
        impute (bool): fill missing observations with posterior means so the data stay closed downwards.
        x_ind (tensor, optional): inducing-point locations; enables the sparse (VFE) path.
        sparse_method (str): "vfe" (default, what the reference uses), "fitc" or "dtc".
    """

    def __init__(self, replace=False, impute=False, x_ind=None, sparse_method="vfe"):
        self.replace = replace
        self.impute = impute
        self.layers = []
        self.sparse = x_ind is not None
        self.x_ind = x_ind
        if sparse_method not in ("vfe", "fitc", "dtc"):
            raise ValueError('sparse_method must be "vfe", "fitc" or "dtc"')
        self.sparse_method = sparse_method  # (an addition: stheno's PseudoObsVFE / FITC / DTC; the reference always uses VFE)

    def copy(self):
        return GPAR(replace=self.replace, impute=self.impute, x_ind=self.x_ind, sparse_method=self.sparse_method)

    def add_layer(self, model_constructor):
        out = self.copy()
        out.layers = self.layers + [model_constructor]
        return out

    # ---- conversions ---------------------------------------------------------------------------
    @staticmethod
    def _prep(x, y, w):
        eng = get_engine()
        x = eng.tensor(x)
        if x.dim() == 1:
            x = x[:, None]
        if not isinstance(y, dict):
            # The NaN pattern of y decides every mask below.  Taken on the host ONCE (free when y arrives as numpy, one small
            # copy when it lives on the GPU), it lets per_output hand out index tensors instead of boolean masks: boolean
            # indexing on the device synchronises - about ten times per layer in the dependent regimes, each time draining the
            # previous layer's factorisation before the host may prepare the next (a 4-layer logpdf at n = 2000 with 10 %
            # missing: 4.5 ms of GPU work in 7.1 ms).
            host_nan, y_given = None, y
            if _is_torch(y):   # (a pattern attached by the caller, or remembered on this very tensor object: see _device_nan_pattern)
                host_nan = _attached_nan_pattern(y)
                if host_nan is None:
                    cached = getattr(y, "_gpar_nan", None)
                    if cached is not None and cached[0] == y._version and cached[1].shape == tuple(y.shape):
                        host_nan = cached[1]
            if isinstance(y, np.ndarray):
                host_nan = np.isnan(y)
            y = eng.tensor(y)
            w = eng.tensor(w)
            if _is_torch(y) and y.is_cuda and y.dim() == 2 and host_masks():
                if host_nan is None:
                    host_nan = _device_nan_pattern(y)
                    if _is_torch(y_given) and y_given.is_cuda:
                        try:
                            y_given._gpar_nan = (y_given._version, host_nan)
                        except (AttributeError, RuntimeError):
                            pass
                y._host_nan, y._host_nan_version = (host_nan if host_nan.ndim == 2 else None), y._version
        return x, y, w

    def _prep_ind(self, x_ind):
        if x_ind is None:
            return None
        t = get_engine().tensor(x_ind)
        return t[:, None] if t.dim() == 1 else t

    # ---- conditioning ----------------------------------------------------------------------------
    def __or__(self, x_y_w):
        """Posterior GPAR given data (x, y, w)."""
        return _retry_unfused(lambda: self._condition(x_y_w), self.layers, x_y_w)

    def _condition(self, x_y_w):
        x, y, w = self._prep(*x_y_w)
        x_ind = self._prep_ind(self.x_ind)
        post = self.copy()
        items = list(per_output(y, w, keep=self.impute))
        eng = get_engine()
        # Observed data only: no layer needs another's posterior, so the factorisations (otherwise done lazily, one
        # after the other, when the posterior is first used) are issued now on alternating streams.
        pipe = eng.pipeline(rows=int(x.shape[0])) if self._independent(items) else None
        # ... or, when they are small, together in lock-step
        lockstep = pipe is not None and self._same_rows(items) and hasattr(eng, "factor_dense_batch") and 0 < int(x.shape[0]) <= eng.batch_rows()
        pending = []
        with eng.defer_checks(), _joining(pipe):  # streams are joined BEFORE the deferred info words are read
            for stage, (is_last, ((yi, wi, mask), model)) in enumerate(last(zip(items, self.layers))):
                complete = isinstance(mask, slice)
                x = x[mask]
                f, noise = model()
                if pipe is not None and lockstep and not _differentiable(f, noise):
                    obs = self._obs(x, x_ind, yi, wi, f, noise, complete=True)
                    pending.append(obs)
                elif pipe is not None and not _differentiable(f, noise):
                    with pipe.stage(stage, x, yi, wi):
                        obs = self._obs(x, x_ind, yi, wi, f, noise, complete=True)
                        obs.factor()
                else:
                    obs = self._obs(x, x_ind, yi, wi, f, noise, complete=complete)
                post.layers.append(construct_model(f | obs, noise))
                if not is_last:
                    x, x_ind = self._update_inputs(x, x_ind, yi, f, obs, complete=complete)
            if pipe is not None:
                pipe.join()
            _lockstep_factors(eng, pending)
        return post

    # ---- log marginal likelihood -------------------------------------------------------------------
    def logpdf(self, x, y, w, only_last_layer=False, sample_missing=False, return_inputs=False, x_ind=None, outputs=None):
        """Sum over layers of log N(y_i; 0, K_i([x, y_<i]) + noise_i / w_i) (the VFE bound with inducing points).

        `outputs` restricts the layers visited, `x_ind` resumes a computation and `return_inputs` returns the
        design matrix (and inducing inputs) reached after the last visited layer instead of the value — the
        three together let `fit` precompute the inputs of a layer once (reference: model.py:178-243)."""
        return _retry_unfused(lambda: self._logpdf(x, y, w, only_last_layer, sample_missing, return_inputs, x_ind, outputs),
                              self.layers, (x, y, w, x_ind))

    def _logpdf(self, x, y, w, only_last_layer, sample_missing, return_inputs, x_ind, outputs):
        x, y, w = self._prep(x, y, w)
        x_ind = self._prep_ind(self.x_ind if x_ind is None else x_ind)
        total = torch.zeros((), dtype=torch.float64)
        items = list(per_output(y, w, keep=self.impute or sample_missing))
        eng = get_engine()
        # layers that do not feed one another (observed data only) are spread over alternating streams
        pipe = eng.pipeline(rows=int(x.shape[0])) if self._independent(items) and not return_inputs else None
        values, stage = [], 0
        # ... or, when they are small enough for a factorisation to be one latency-bound chain, factored together in lock-step
        lockstep = pipe is not None and self._same_rows(items) and hasattr(eng, "logpdf_dense_batch") and 0 < int(x.shape[0]) <= eng.batch_rows()
        pending = []
        # ... and then, when y and w are whole device matrices and every layer is a prior process outside autograd, through ONE
        # library call (_lockstep_total)
        onecall = (lockstep and outputs is None and hasattr(eng, "logpdf_lockstep") and _is_torch(y) and y.is_cuda and y.dim() == 2
                   and _is_torch(w) and w.shape == y.shape and getattr(eng, "cholesky_retry_factor", 1.0) <= 1.0 and one_call_enabled())
        if onecall:
            for model in self.layers[:len(items)]:
                f, noise = model()
                if f.is_posterior or _differentiable(f, noise):
                    onecall = False
                    break
        fast, visited = [], 0
        with eng.defer_checks(), _joining(pipe):  # streams are joined BEFORE the deferred info words are read
            for is_last, ((yi, wi, mask), model) in last(zip(items, self.layers), select=outputs):
                complete = isinstance(mask, slice)
                x = x[mask]
                f, noise = model()
                if pipe is not None and not onecall and _differentiable(f, noise):   # (the one-call route has checked every layer)
                    pipe.join()
                    pipe = None  # an objective under autograd: keep everything on the caller's stream
                if pipe is not None and onecall:
                    # the one-call route: nothing is built per layer; the design matrix [x, y_<i] is a prefix of one widest matrix
                    if not only_last_layer or is_last:
                        fast.append((visited, f, noise))
                    visited += 1
                    continue
                if pipe is not None and lockstep:
                    if not only_last_layer or is_last:
                        obs = self._obs(x, x_ind, yi, wi, f, noise, complete=True)
                        obs.transient = True
                        pending.append((f, obs))
                    if not is_last:
                        x = torch.cat([x, yi], dim=1)
                    continue
                if pipe is not None:
                    if not only_last_layer or is_last:
                        with pipe.stage(stage, x, yi, wi):
                            obs = self._obs(x, x_ind, yi, wi, f, noise, complete=True)
                            obs.transient = True   # nobody conditions on this layer: only its value is wanted
                            values.append(f.measure.logpdf(obs))
                        stage += 1
                    if not is_last:
                        x = torch.cat([x, yi], dim=1)
                    continue
                obs = self._obs(x, x_ind, yi, wi, f, noise, complete=complete)
                if not only_last_layer or is_last:
                    total = total + f.measure.logpdf(obs)
                if not is_last:
                    if sample_missing and not complete:
                        missing = torch.isnan(yi[:, 0])
                        if bool(missing.any()):
                            f_post = f | obs
                            drawn = f_post(x[missing], self._noise_over(noise, wi[missing])).sample()
                            yi = merge(yi, drawn, missing)
                    x, x_ind = self._update_inputs(x, x_ind, yi, f, obs, complete=complete)
            if pipe is not None:
                pipe.join()
            if fast:
                got = _lockstep_total(eng, x, y, w, fast)
                if got is None:   # no room for the batch: layer by layer
                    xw = torch.cat([x, y[:, :visited]], dim=1)
                    for j, fj, nj in fast:
                        obs = self._obs(xw[:, :int(x.shape[1]) + j], x_ind, y[:, j:j + 1], w[:, j], fj, nj, complete=True)
                        obs.transient = True
                        pending.append((fj, obs))
                else:
                    values.extend(got)
            if pending:
                values.extend(_lockstep_values(eng, pending))
            for v in values:
                # (the first term replaces the host-side zero it would be added to: 0 + v is v, and the addition is a launch)
                total = v if (total.device.type == "cpu" and not total.requires_grad and total.dim() == 0 and float(total) == 0.0 and _is_torch(v)) else total + v
        if return_inputs:
            return x, x_ind
        return total.cpu() if total.is_cuda and not total.requires_grad else total

    def _independent(self, items):
        """No layer needs anything a previous layer computes: no `replace`, no inducing points, and nothing to impute - complete
        data (slice masks), or rows dropped per layer with every kept row observed (`impute=False` with missing data; known
        without a synchronisation when the NaN pattern was planned on the host) - the design matrix of layer i is just [x, y_<i]."""
        def nothing_missing(item):
            yi, _, mask = item
            if isinstance(mask, slice) or getattr(yi, "_n_missing", None) == 0:
                return True
            return not mask.is_cuda and mask.dtype == torch.bool and bool(mask.all())   # an all-True mask on the CPU (free to test there)

        return not self.replace and not self.sparse and len(items) > 1 and all(nothing_missing(item) for item in items)

    @staticmethod
    def _same_rows(items):
        """Every layer sees all rows (what lock-step batches need: one matrix size)."""
        return all(isinstance(mask, slice) or (not mask.is_cuda and mask.dtype == torch.bool and bool(mask.all())) for _, _, mask in items)

    # ---- sampling ----------------------------------------------------------------------------------
    def sample(self, x, w, latent=False):
        """One ancestral sample, n x p (reference: model.py:245-277).  With `latent` the noise-free function
        values are returned while the noisy values are what is fed to the next layer."""
        return _retry_unfused(lambda: self._sample(x, w, latent), self.layers, (x, w), rewind=True)

    def _sample(self, x, w, latent):
        eng = get_engine()
        x = eng.tensor(x)
        if x.dim() == 1:
            x = x[:, None]
        w = eng.tensor(w)
        columns = []
        x_ind = self._prep_ind(self.x_ind)
        for i, (is_last, model) in enumerate(last(self.layers)):
            f, noise = model()
            if latent:
                f_sample = f(x).sample()
                std = torch.sqrt(self._noise_over(noise, w[:, i : i + 1]))
                y_sample = f_sample + std * eng.randn(f_sample.shape[0], 1)
                columns.append(f_sample)
            else:
                y_sample = f(x, self._noise_over(noise, w[:, i])).sample()
                columns.append(y_sample)
            if not is_last:
                x, x_ind = self._update_inputs(x, x_ind, y_sample, f, None)
        if not columns:
            return torch.zeros(x.shape[0], 0, dtype=torch.float64, device=x.device)
        return torch.cat(columns, dim=1)

    def sample_many(self, x, w, num_samples, latent=False, marginal=False):
        """`num_samples` independent ancestral samples (the loop of reference regression.py:559-563), computed layer by
        layer for all samples at once.  While every sample still sees the same design matrix (always at layer 0;
        at every layer when `replace` feeds posterior means forward) one factorisation serves all draws; once the
        inputs differ per sample, the per-sample cross-covariances are stacked into one triangular solve.

        `marginal` (not in the reference; off by default): within a layer every point is drawn from its own marginal
        N(mean_j, var_j) instead of the joint law over the n* points.  Per-point predictive statistics (what `predict`
        reports: means and marginal percentiles) have the same distribution; the joint law of one sample across points
        does not, so `sample` keeps the joint sampler.  No n* x n* covariance is built or factored."""
        if num_samples == 1 and not marginal:
            return [self.sample(x, w, latent=latent)]
        return _retry_unfused(lambda: self._sample_many(x, w, num_samples, latent, marginal), self.layers, (x, w), rewind=True)

    def _sample_many(self, x, w, num_samples, latent, marginal):
        eng = get_engine()
        x = eng.tensor(x)
        if x.dim() == 1:
            x = x[:, None]
        w = eng.tensor(w)
        S, ns = num_samples, x.shape[0]
        x_ind = self._prep_ind(self.x_ind)
        shared, xs = True, None
        columns = []  # per layer: n* x S
        for i, (is_last, model) in enumerate(last(self.layers)):
            f, noise = model()
            obs_noise = None if latent else self._noise_over(noise, w[:, i])
            if marginal:
                if shared:
                    mean, var = f.marginal_moments(x, obs_noise)
                    draws = mean + torch.sqrt(torch.clamp(var + eng.epsilon, min=0.0))[:, None] * eng.randn(ns, S)
                else:
                    draws = f.marginal_sample_batch(xs, obs_noise)
            else:
                draws = f(x, obs_noise).sample(num=S) if shared else f.sample_batch(xs, obs_noise)
            columns.append(draws)
            if is_last:
                break
            if latent:
                fed = draws + torch.sqrt(self._noise_over(noise, w[:, i : i + 1])) * eng.randn(ns, S)
            else:
                fed = draws
            if self.sparse:
                x_ind = torch.cat([x_ind, f.mean(x_ind)], dim=1)
            if self.replace:
                # the sampled values are replaced by the (posterior) mean at the current inputs
                if shared:
                    x = torch.cat([x, f.mean(x)], dim=1)
                else:
                    xs = xs.with_columns(torch.cat(list(f.mean_batch(xs)), dim=0))
            elif shared:
                xs = Stacked.repeat(x, S).with_columns(fed)   # from here on every sample has a design matrix of its own
                shared = False
            else:
                xs = xs.with_columns(fed)
        # one stack (n* x S x p) and S views of it, instead of S stacks of p columns each
        return list(torch.stack(columns, dim=2).permute(1, 0, 2).unbind(0))

    def moments(self, x, w, latent=False):
        """(mean, variance), n* x p each, of the samples `sample` would draw at x - in closed form, which exists when `replace`
        feeds the layers' MEANS forward (reference model.py:245-277 with `_update_inputs` :291-322 and obs = None): the inputs
        of every layer are then deterministic, x_{i+1} = [x_i, mean_i(x_i)], and a draw of layer i at point j is
        N(mean_i(x_i)_j, var_i(x_i)_j [+ noise_i / w_ij]).  What the Monte-Carlo mean and spread of `predict` converge to,
        without a sample, an n* x n* covariance or a factorisation of one (an addition: the reference only samples)."""
        if not self.replace:
            raise ValueError("closed-form predictive moments need replace=True: otherwise samples, not means, are fed forward")
        return _retry_unfused(lambda: self._moments(x, w, latent), self.layers, (x, w), rewind=True)

    def _moments(self, x, w, latent):
        eng = get_engine()
        x = eng.tensor(x)
        if x.dim() == 1:
            x = x[:, None]
        w = eng.tensor(w)
        means, variances = [], []
        for i, (is_last, model) in enumerate(last(self.layers)):
            f, noise = model()
            mean, var = f.marginal_moments(x, None if latent else self._noise_over(noise, w[:, i]))
            means.append(mean.reshape(-1, 1))
            variances.append(var.reshape(-1, 1))
            if not is_last:
                x = torch.cat([x, mean.reshape(-1, 1)], dim=1)
        return torch.cat(means, dim=1), torch.cat(variances, dim=1)

    # ---- helpers -----------------------------------------------------------------------------------
    @staticmethod
    def _noise_over(noise, w):
        """noise / w on w's device (noise is a Python float or a CPU 0-d tensor)."""
        # NB: `python_float / tensor` is evaluated by torch as `tensor.reciprocal() * float`, which is not the
        # correctly rounded quotient; the reference divides two tensors, so do the same.  The numerator is made on the
        # device by a fill (a kernel argument): copying a CPU scalar over would be a SYNCHRONOUS host-to-device transfer
        # queued behind everything on the stream - one hidden host sync per layer (1.3 ms each at n = 4096).
        value = float(noise.detach()) if _is_torch(noise) else float(noise)
        return torch.true_divide(torch.full((), value, dtype=torch.float64, device=w.device), w)

    def _obs(self, x, x_ind, y, w, f, noise, complete=False):
        eng = get_engine()
        n_missing, keep_rows = getattr(y, "_n_missing", None), getattr(y, "_obs_idx", None)   # (the plan sits on the tensor per_output yielded)
        x, y, w = eng.tensor(x), eng.tensor(y), eng.tensor(w)
        if not complete:  # `complete`: the caller knows no observation is missing (saves a host sync per layer)
            if n_missing is None:
                available = ~torch.isnan(y[:, 0])
                x, y, w = x[available], y[available], w[available]
            elif n_missing:   # (planned on the host, per_output: index tensors, no synchronisation)
                x, y, w = x.index_select(0, keep_rows), y.index_select(0, keep_rows), w.index_select(0, keep_rows)
        if self.sparse:
            cls = {"vfe": PseudoObs, "fitc": PseudoObsFITC, "dtc": PseudoObsDTC}[self.sparse_method]
            return cls(f(x_ind), f(x, self._noise_arg(noise, w)), y)
        return Obs(f(x, self._noise_arg(noise, w)), y)

    @staticmethod
    def _noise_arg(noise, w):
        """noise / w, keeping the autograd graph of `noise` when it has one (used by the training objective)."""
        if _is_torch(noise) and noise.requires_grad:
            return noise / w if noise.device == w.device or noise.dim() == 0 else noise.to(w.device) / w
        return GPAR._noise_over(noise, w)

    def _update_inputs(self, x, x_ind, y, f, obs, complete=False):
        """Append output column y to the design matrix (and the estimated output to the inducing inputs)."""
        eng = get_engine()
        # (set by per_output when the NaN pattern is known on the host; read before the conversion below makes a new tensor object)
        n_missing, obs_rows, miss_rows = getattr(y, "_n_missing", None), getattr(y, "_obs_idx", None), getattr(y, "_miss_idx", None)
        x, y = eng.tensor(x), eng.tensor(y)
        x_ind = None if x_ind is None else eng.tensor(x_ind)
        if complete and not self.sparse and not self.replace:
            return torch.cat([x, y], dim=1), x_ind  # nothing to estimate: observed column, no host sync
        available = ~torch.isnan(y[:, 0]) if n_missing is None else None
        post = (f | obs) if obs else None

        def estimate(x_):
            return post.mean(x_) if post is not None else f.mean(x_)

        if self.sparse:
            x_ind = torch.cat([x_ind, estimate(x_ind)], dim=1)
        if (self.impute and self.replace) or (self.replace and complete):
            y = estimate(x)   # (complete data: every row is observed, so "replace the observed ones" is all of them)
        elif not complete and n_missing is not None:
            # planned on the host: index tensors, no `.any()`, no boolean indexing
            n_rows = int(y.shape[0])
            if self.impute and n_missing:
                y = _merge_rows(y, estimate(x.index_select(0, miss_rows)), miss_rows)
            if self.replace and n_missing < n_rows:
                y = _merge_rows(y, estimate(x.index_select(0, obs_rows)), obs_rows) if n_missing else estimate(x)
        elif not complete:
            # (`complete` is known from the masks: with it there is nothing to impute, and no `.any()` - a host sync per layer,
            # after which the host cannot prepare layer i + 1 while the GPU works on layer i: 0.3-0.5 ms per layer at C4)
            if self.impute and bool((~available).any()):
                y = merge(y, estimate(x[~available]), ~available)
            if self.replace and bool(available.any()):
                y = merge(y, estimate(x[available]), available)
        return torch.cat([x, y], dim=1), x_ind
