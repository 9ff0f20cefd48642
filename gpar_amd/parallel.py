"""Layer-parallel GPAR over the GPUs of one node: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on the MI355X box, "gloo" in the CPU tests).

The p autoregressive layers are the units of work: layer i is owned by rank i mod G.  What has to travel between
ranks is decided by the model, exactly as in /root/reference/gpar/model.py:291-322:

  * `replace=False`, nothing to impute, no inducing points: layer i's design matrix is `[x, observed y_<i]`, which
    every rank already holds -> layers are independent, the data path has NO collective; only the p scalar
    log-likelihoods are summed (one 8-byte all-reduce).
  * otherwise (posterior means are fed forward): the owner of layer i computes the new input column (and the new
    inducing-input column) and broadcasts it — n x 8 bytes per layer, latency-bound — before layer i+1 can start.

With `markov=k` a forwarded column is read by the next k layers only (reference regression.py:49-59: the kernel of layer j
selects the last k outputs), so it is SENT to the owners of those layers alone (point-to-point) instead of broadcast; ranks
that never read a column hold NaN in its place (a violation would be loud).

`fit(fix=True)` trains each layer on its owner (the optimiser only touches names "{i}/*", reference
regression.py:453-454) and then broadcasts the trained latent variables so every rank holds the same `Vars`.
`fit(fix=False)` (the joint objective, reference regression.py:447-456) shards the sum over layers the same way: every rank
evaluates its layers' terms and their gradients, one all-reduce adds values and gradient vectors (the slices "{i}/*" are
disjoint, the tied "0/input/scales" receives every layer's contribution), and ONE L-BFGS-B driver on rank 0 broadcasts the
iterates.  Where layer inputs are posterior means of earlier layers (`replace`, imputation, inducing points) training is
REPLICATED: every rank runs the serial `fit` (`sharded_fit` returns which of the three it did).
`predict` conditions layer-parallel (`sharded_condition`), splits the Monte-Carlo samples across ranks, all-gathers the
device-resident sample stacks once and reduces them with `gpar_sample_stats` (`sharded_predict`).
"""
import numpy as np
import torch
import torch.distributed as dist

from .engine import get_engine
from .model import host_masks, last, per_output

__all__ = ["world", "sharded_logpdf", "sharded_fit", "sharded_condition", "sharded_sample", "sharded_predict", "forward_plan"]


def world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _needs_estimate(gpar, yi, complete):
    if gpar.sparse or gpar.replace:
        return True
    if complete or not gpar.impute:
        return False
    return bool(torch.isnan(yi[:, 0]).any())


def sharded_logpdf(gpar, x, y, w, group=None, timing=None):
    """`GPAR.logpdf(x, y, w)` with the layers divided over the ranks of `group`; every rank returns the total.
    `timing` (a dict, optional) accumulates under "busy_s" the wall-clock this rank spent on its own layers, i.e. up to
    the collective, and under "collective_s" the time in the 8-byte all-reduce - which includes the wait for the slowest rank (what
    makes a multi-GPU run interpretable without a second one: the slowest rank's busy time bounds the step, a fast rank's
    collective time is the imbalance)."""
    import time

    t_start = time.perf_counter()
    rank, size = world(group)
    eng = get_engine()
    x, y, w = gpar._prep(x, y, w)
    x_ind = gpar._prep_ind(gpar.x_ind)
    local = torch.zeros((), dtype=torch.float64)
    with eng.defer_checks():
        local, x, x_ind = _sharded_layers(gpar, x, y, w, x_ind, rank, size, group, local)
    if local.is_cuda:
        local = local.cpu()
    t_busy = time.perf_counter()
    if timing is not None:
        timing["busy_s"] = timing.get("busy_s", 0.0) + (t_busy - t_start)
    if size > 1:
        buf = local.detach().to(device=eng.device, dtype=torch.float64).reshape(1).clone()
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        local = buf[0].cpu()
        if timing is not None:   # (includes the wait for the slowest rank: busy + collective = the step on every rank)
            timing["collective_s"] = timing.get("collective_s", 0.0) + (time.perf_counter() - t_busy)
    return local


def _columns_read(kernel, width):
    """Design-matrix columns a layer kernel reads (a factor without a selection reads all `width` of them)."""
    cols = set()
    for term in kernel.terms:
        for factor in term.factors:
            cols.update(range(width) if factor.cols is None else (int(c) for c in factor.cols))
    return cols


def forward_plan(gpar, m, size):
    """needs[i] = ranks that read output column i (design-matrix column m + i): the owners (j mod size) of the layers j > i whose
    kernel selects it - all later layers without `markov`, the next k with `markov=k` (reference regression.py:49-59).  The
    owner of layer j also evaluates layer j's posterior mean when it forwards column j, through the same kernel."""
    import os

    count = len(gpar.layers)
    if os.environ.get("GPAR_MARKOV_SENDS", "1") == "0":
        return [set(range(size)) for _ in range(count)]
    needs = [set() for _ in range(count)]
    for j, model in enumerate(gpar.layers):
        f, _ = model()
        for c in _columns_read(f.kernel, m + j):
            if m <= c < m + j:
                needs[c - m].add(j % size)
    return needs


def _forward(col, ind_col, mine, owner, receivers, rank, size, group):
    """The forwarded column (and its inducing-input counterpart) from `owner` to `receivers` (group ranks): one broadcast when
    every other rank reads it, point-to-point sends otherwise; ranks that do not read it get NaN."""
    others = set(range(size)) - {owner}
    targets = set(receivers) - {owner}
    if not targets:
        if not mine:
            col.fill_(float("nan"))
            if ind_col is not None:
                ind_col.fill_(float("nan"))
        return
    if targets == others:
        dist.broadcast(col, src=_global_rank(owner, group), group=group)
        if ind_col is not None:
            dist.broadcast(ind_col, src=_global_rank(owner, group), group=group)
        return
    # one message per receiver: [column; inducing column]
    packed = col.reshape(-1) if ind_col is None else torch.cat([col.reshape(-1), ind_col.reshape(-1)])
    if mine:
        packed = packed.contiguous()
        for work in [dist.isend(packed, dst=_global_rank(r, group), group=group) for r in sorted(targets)]:
            work.wait()
    elif rank in targets:
        buf = torch.empty_like(packed)
        dist.recv(buf, src=_global_rank(owner, group), group=group)
        col.copy_(buf[: col.numel()].reshape(col.shape))
        if ind_col is not None:
            ind_col.copy_(buf[col.numel():].reshape(ind_col.shape))
    else:
        col.fill_(float("nan"))
        if ind_col is not None:
            ind_col.fill_(float("nan"))


def _sharded_layers(gpar, x, y, w, x_ind, rank, size, group, local):
    from .model import _differentiable, _joining, _lockstep_values

    items = list(per_output(y, w, keep=gpar.impute))
    eng = get_engine()
    needs = forward_plan(gpar, int(x.shape[1]), size) if size > 1 else None
    # this rank's layers alternate over two streams when no layer feeds another (see HipEngine.pipeline) ...
    pipe = eng.pipeline(rows=int(x.shape[0])) if gpar._independent(items) else None
    # ... or are factored together in lock-step
    lockstep = pipe is not None and gpar._same_rows(items) and hasattr(eng, "logpdf_dense_batch") and 0 < int(x.shape[0]) <= eng.batch_rows()
    pending = []
    values, stage = [], 0
    # ... through ONE library call when y and w are whole device matrices and the layers are prior processes (model._lockstep_total)
    from .model import _is_torch, _lockstep_total, one_call_enabled

    onecall = (lockstep and hasattr(eng, "logpdf_lockstep") and _is_torch(y) and y.is_cuda and y.dim() == 2 and _is_torch(w)
               and w.shape == y.shape and getattr(eng, "cholesky_retry_factor", 1.0) <= 1.0 and one_call_enabled())
    if onecall:
        for i, model in enumerate(gpar.layers[:len(items)]):
            if i % size == rank:
                f, noise = model()
                if f.is_posterior or _differentiable(f, noise):
                    onecall = False
                    break
    fast, x0 = [], x
    with _joining(pipe):
        for i, (is_last, ((yi, wi, mask), model)) in enumerate(last(zip(items, gpar.layers))):
            complete = isinstance(mask, slice)
            x = x[mask]
            mine = (i % size) == rank
            f = obs = None
            if onecall:   # (independent layers, complete data: nothing is forwarded, no design matrix is formed per layer)
                if mine:
                    f, noise = model()
                    fast.append((i, f, noise))
                continue
            if mine:
                f, noise = model()
                if pipe is not None and _differentiable(f, noise):
                    pipe.join()
                    pipe = None
                if pipe is not None and lockstep:
                    obs = gpar._obs(x, x_ind, yi, wi, f, noise, complete=True)
                    obs.transient = True
                    pending.append((f, obs))
                elif pipe is not None:
                    with pipe.stage(stage, x, yi, wi):
                        values.append(f.measure.logpdf(gpar._obs(x, x_ind, yi, wi, f, noise, complete=True)))
                    stage += 1
                else:
                    obs = gpar._obs(x, x_ind, yi, wi, f, noise, complete=complete)
                    local = local + f.measure.logpdf(obs)
            if is_last:
                break
            if not _needs_estimate(gpar, yi, complete):
                x = torch.cat([x, yi], dim=1)  # observed data: already on every rank
                continue
            # dependent chain: the owner computes the forwarded column(s), everyone else receives them
            n_i = x.shape[0]
            col = torch.empty(n_i, 1, dtype=torch.float64, device=x.device)
            ind_col = None if x_ind is None else torch.empty(x_ind.shape[0], 1, dtype=torch.float64, device=x.device)
            if mine:
                x_new, x_ind_new = gpar._update_inputs(x, x_ind, yi, f, obs, complete=complete)
                col.copy_(x_new[:, -1:])
                if ind_col is not None:
                    ind_col.copy_(x_ind_new[:, -1:])
            if size > 1:
                _forward(col, ind_col, mine, i % size, needs[i], rank, size, group)
            x = torch.cat([x, col], dim=1)
            if ind_col is not None:
                x_ind = torch.cat([x_ind, ind_col], dim=1)
    if pipe is not None:
        pipe.join()
    if fast:
        got = _lockstep_total(eng, x0, y, w, fast)
        if got is None:   # no room for the batch: layer by layer
            xw = torch.cat([x0, y[:, :len(items) - 1]], dim=1)
            for j, fj, nj in fast:
                obs = gpar._obs(xw[:, :int(x0.shape[1]) + j], x_ind, y[:, j:j + 1], w[:, j], fj, nj, complete=True)
                obs.transient = True
                pending.append((fj, obs))
        else:
            values.extend(got)
    if pending:
        values.extend(_lockstep_values(eng, pending))
    for v in values:
        local = local + v
    if onecall:
        x = torch.cat([x0, y[:, :len(items) - 1]], dim=1) if len(items) > 1 else x0
    return local, x, x_ind


def _global_rank(group_rank, group):
    if group is None:
        return group_rank
    return dist.get_global_rank(group, group_rank)


def layers_train_independently(reg, y):
    """With `fix=True`, is the training of layer pi independent of the training of the other layers?  Its inputs must be
    data only (no `replace`, no inducing points, nothing to impute) and it must not share a hyper-parameter with
    another layer (`scale_tie` reads "0/input/scales", which only layer 0 trains)."""
    return inputs_are_data(reg, y) and not reg.model_config.get("scale_tie", False)


def inputs_are_data(reg, y):
    """Every layer's design matrix is [x, observed y_<i]: nothing a layer computes is fed to a later one (no `replace`, no
    inducing points, nothing to impute), so the terms of the log-likelihood are separate functions of the hyper-parameters."""
    return not (reg.replace or reg.sparse or (reg.impute and bool(torch.isnan(y).any())))


def _tag_host_nan(reg, y_dev):
    """The NaN pattern of the (host-resident) training outputs, attached for per_output: masks planned on the host (GPAR._prep)."""
    if y_dev.is_cuda and host_masks():
        y_dev._host_nan, y_dev._host_nan_version = torch.isnan(reg.y).numpy(), y_dev._version


def sharded_fit(reg, x, y, w=None, group=None, fix=True, **kw_args):
    """`GPARRegressor.fit(x, y, w, fix=fix)` over the ranks of `group`; every rank ends up with the same hyper-parameters.
    Returns what it did:
      "layer-parallel"  fix=True, inputs are data, no tied scales: layer pi trained on rank pi mod G, results broadcast;
      "joint-sharded"   fix=False, inputs are data: the reference's p successive joint optimisations (layers 0..pi together,
                        regression.py:447-456) with the sum over layers divided over the ranks - see `_minimise_sharded`;
      "replicated"      layer inputs are posterior means of earlier layers (or, with fix=True, scales are tied): the chain is
                        sequential, every rank runs the serial `fit`."""
    from .optimise import minimise_l_bfgs_b
    from .regression import _construct_gpar

    rank, size = world(group)
    eng = get_engine()
    reg.condition(x, y, w)
    x_dev, y_dev, w_dev = eng.tensor(reg.x), eng.tensor(reg.y), eng.tensor(reg.w)
    _tag_host_nan(reg, y_dev)
    if not inputs_are_data(reg, y_dev) or (fix and not layers_train_independently(reg, y_dev)):
        # inputs of layer pi depend on the trained layers < pi: the chain is sequential; train replicated
        reg.fit(x, y, w, fix=fix, **kw_args)
        return "replicated"
    y_cached = {k: list(per_output(y_dev, w_dev, keep=k)) for k in [True, False]}
    # instantiate every variable on every rank (lazy creation, reference regression.py:92-180)
    with torch.no_grad():
        _construct_gpar(reg, reg.vs, reg.m, reg.p).logpdf(x_dev[:2], y_dev[:2], w_dev[:2])
    # the design matrix of an owned layer, [x, y_<pi] with its rows: data only, so it is formed once
    fixed = {}
    for pi in range(reg.p):
        if pi % size == rank:
            gpar = _construct_gpar(reg, reg.vs, reg.m, pi + 1)
            fixed[pi] = gpar.logpdf(x_dev, y_cached, None, only_last_layer=True, outputs=list(range(pi)), return_inputs=True)

    def term(vs, pi):
        g = _construct_gpar(reg, vs, reg.m, pi + 1)
        return -g.logpdf(fixed[pi][0], y_cached, None, only_last_layer=True, outputs=[pi], x_ind=fixed[pi][1])

    if not fix:
        for pi in range(reg.p):
            owned = [i for i in range(pi + 1) if i % size == rank]

            def local_objective(vs, owned=owned):
                total = torch.zeros((), dtype=torch.float64)
                for i in owned:
                    total = total + term(vs, i)
                return total

            _minimise_sharded(local_objective, reg.vs, [f"{i}/*" for i in range(pi + 1)], group, **kw_args)
        return "joint-sharded"
    # this rank's layers through the regressor's own layer-wise training: the prepared objective (gpar_amd/fastfit.py) and the
    # worker streams where they apply, the general route otherwise - what the single-process `fit` runs for the same layers
    reg._train(sorted(fixed), fix=True, **kw_args)
    if size > 1:
        for pi in range(reg.p):
            names = reg.vs.match([f"{pi}/*"])
            if not names:
                continue
            vec = torch.as_tensor(reg.vs.get_vector(names), dtype=torch.float64).to(eng.device)
            dist.broadcast(vec, src=_global_rank(pi % size, group), group=group)
            reg.vs.set_vector(vec.cpu().numpy(), names)
    return "layer-parallel"


def _minimise_sharded(local_objective, vs, patterns, group, iters=1000, f_calls=10000, trace=False):
    """L-BFGS-B over the variables matching `patterns` of an objective that is a SUM of per-rank terms: `local_objective(vs)`
    is this rank's share.  One evaluation = every rank evaluates its share and back-propagates, then ONE all-reduce adds
    [value, gradient] over the ranks (a variable only this rank's layers read gets zeros from the others; a tied variable gets
    every layer's contribution).  The optimiser itself runs on rank 0 only and broadcasts each iterate - [1, x] to evaluate,
    [0, x_opt] to finish - so the ranks cannot drift apart, whatever their arithmetic.  Returns the final objective value."""
    import logging

    import scipy.optimize

    from . import optimise
    from .engine import NotPositiveDefiniteError

    rank, size = world(group)
    eng = get_engine()
    names = vs.match(patterns)
    latents = vs.get_vars(*names)
    if not latents:
        raise ValueError("no variables to optimise")
    x0 = vs.get_vector(names)
    dim = x0.size

    def evaluate(xvec):
        optimise._evaluations += 1
        vs.set_vector(xvec, names)
        previous = [t.requires_grad for t in latents]
        for t in latents:
            t.requires_grad_(True)
            t.grad = None
        fatal = None
        try:
            value = local_objective(vs)
            if value.requires_grad:
                value.backward()
            grad = np.concatenate([(t.grad if t.grad is not None else torch.zeros_like(t)).detach().numpy().reshape(-1) for t in latents])
            val = float(value.detach())
        except (NotPositiveDefiniteError, ArithmeticError) as e:  # as varz: report NaN and let the line search back off
            logging.getLogger(__name__).warning("objective evaluation failed (%s); returning NaN", e)
            val, grad = np.nan, np.zeros(dim)
        except Exception as e:  # noqa: BLE001 - anything else (a hand-off timeout re-raised under autograd, out of memory, a failed
            # point-to-point transfer) must not leave the other ranks waiting in the all-reduce below: this rank takes part in it,
            # with a failure flag every rank sees, and all of them raise afterwards
            fatal = e
            val, grad = np.nan, np.zeros(dim)
        finally:
            for t, r in zip(latents, previous):
                t.requires_grad_(r)
                t.grad = None
        buf = torch.as_tensor(np.concatenate([[val], grad, [0.0 if fatal is None else 1.0]]), dtype=torch.float64).to(eng.device)
        if size > 1:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        out = buf.cpu().numpy()
        if out[-1] > 0.0:
            if fatal is not None:
                raise fatal
            raise RuntimeError(f"the joint objective failed on {int(out[-1])} other rank(s) (their exception is raised there)")
        val, grad = float(out[0]), out[1:-1].copy()
        if not np.isfinite(val):
            val, grad = np.nan, np.zeros(dim)   # some rank's share failed: a failed evaluation for everybody
        if trace and rank == 0:
            print(f"  objective {val:.6e}  |grad| {np.linalg.norm(grad):.3e}")
        return val, grad

    def announce(flag, xvec):
        ctrl = torch.as_tensor(np.concatenate([[flag], np.asarray(xvec, dtype=np.float64)]), dtype=torch.float64).to(eng.device)
        if size > 1:
            dist.broadcast(ctrl, src=_global_rank(0, group), group=group)
        return ctrl.cpu().numpy()

    if rank == 0:
        def fg(xvec):
            announce(1.0, xvec)
            return evaluate(xvec)

        x_opt, val, _ = scipy.optimize.fmin_l_bfgs_b(fg, x0, maxiter=iters, maxfun=f_calls)
        announce(0.0, x_opt)
        vs.set_vector(x_opt, names)
        return val
    val = np.nan   # (never returned as such: scipy's driver evaluates its starting point, so rank 0 announces at least one evaluation)
    while True:
        ctrl = announce(0.0, np.zeros(dim))
        if ctrl[0] == 0.0:
            vs.set_vector(ctrl[1:], names)
            return val
        val, _ = evaluate(ctrl[1:])


def sharded_condition(reg, group=None):
    """The conditioned GPAR of `reg` (as `gpar | (x, y, w)`), with the p training-data factorisations divided over the
    ranks when no layer feeds another: layer i is factored by rank i mod G (a rank's layers pipelined over its streams)
    and the factor buffers - L and the row L^-1 y - are all-gathered in packed lower-triangular form, G layers per
    collective; every rank ends up holding every factor, which is what sample-parallel prediction needs.  In the dependent regimes (imputation,
    `replace`, inducing points) every rank conditions locally, as the chain is sequential anyway."""
    from .model import construct_model
    from .regression import _construct_gpar

    rank, size = world(group)
    eng = get_engine()
    gpar = _construct_gpar(reg, reg.vs, reg.m, reg.p)
    x, y, w = gpar._prep(reg.x, reg.y, reg.w)
    items = list(per_output(y, w, keep=gpar.impute))
    if size == 1 or not gpar._independent(items) or not gpar._same_rows(items):
        # (layers whose rows differ - `impute=False` with missing data - are independent too, but their factors do not share a
        # shape: they are conditioned locally rather than exchanged)
        return gpar | (reg.x, reg.y, reg.w)
    from .engine import joining

    from .model import _lockstep_factors

    post = gpar.copy()
    pipe = eng.pipeline(rows=int(x.shape[0]))
    # this rank's layers: in lock-step when they are small enough, else over its streams
    lockstep = pipe is not None and gpar._same_rows(items) and hasattr(eng, "factor_dense_batch") and 0 < int(x.shape[0]) <= eng.batch_rows()
    factors, mine = [], []
    with eng.defer_checks(), joining(pipe):
        for i, (is_last, ((yi, wi, mask), model)) in enumerate(last(zip(items, gpar.layers))):
            x = x[mask]
            f, noise = model()
            obs = gpar._obs(x, None, yi, wi, f, noise, complete=True)
            if i % size == rank:
                if lockstep:
                    mine.append((i, obs))
                    factors.append(None)
                elif pipe is not None:
                    with pipe.stage(i // size, x, yi, wi):
                        factors.append(obs.factor())
                else:
                    factors.append(obs.factor())
            else:
                factors.append(obs.adopt_factor())
            post.layers.append(construct_model(f | obs, noise))
            if not is_last:
                x = torch.cat([x, yi], dim=1)
        if mine:
            _lockstep_factors(eng, [o for _, o in mine])
            for i, o in mine:
                factors[i] = o.factor()
    # The exchange step: layer i's factor (L and the row L^-1 y: the lower triangle of the (n + 1) x (n + 1) buffer) lives on
    # rank i mod G.  Round k all-gathers layers k G .. k G + G - 1 in PACKED form - (n + 1)(n + 2) / 2 doubles each, half the
    # bytes of the padded square buffers and one collective over all xGMI links per round instead of one broadcast (one
    # root's links) per layer.  The streams were joined above, so the collective is ordered after the factorisations.
    _exchange_factors(eng, factors, rank, size, group)
    return post


def _exchange_factors(eng, factors, rank, size, group):
    count = len(factors)
    for first in range(0, count, size):
        mine = first + rank
        sizes = [f.n + 1 for f in factors[first : first + size]]
        if len(set(sizes)) != 1:
            # ragged layers (sharded_condition only exchanges layers that share their rows; kept for direct callers): one
            # broadcast per layer
            for i in range(first, min(first + size, count)):
                buf = factors[i].A._base if factors[i].A._base is not None else factors[i].A
                dist.broadcast(buf, src=_global_rank(i % size, group), group=group)
            continue
        N = sizes[0]
        length = N * (N + 1) // 2
        send = torch.empty(length, dtype=torch.float64, device=eng.device)
        if mine < count:
            eng.pack_lower(factors[mine].A, send)
        else:
            send.zero_()  # this rank has no layer in the last, incomplete round
        recv = [torch.empty(length, dtype=torch.float64, device=eng.device) for _ in range(size)]
        dist.all_gather(recv, send, group=group)
        for r in range(size):
            i = first + r
            if i < count and r != rank:
                eng.unpack_lower_(recv[r], factors[i].A)


def _sharded_sample_stack(reg, x, w, num_samples, latent, group, marginal=False):
    """(S x n* x p device tensor holding every rank's draws, in rank order; the per-rank counts): the conditioning is
    layer-parallel (`sharded_condition`), each rank draws its share with its own Philox stream, and the device-resident stacks
    are all-gathered ONCE - no sample visits the host."""
    rank, size = world(group)
    eng = get_engine()
    counts = [num_samples // size + (1 if r < num_samples % size else 0) for r in range(size)]
    eng.seed(getattr(eng, "_seed", 0) * 1000003 + rank + 1)
    post = sharded_condition(reg, group) if size > 1 else None
    mine = reg._sample_device(x, w, None, True, max(counts[rank], 1), latent, conditioned=post, marginal=marginal)[: counts[rank]]
    rows = int(np.shape(x)[0])
    local = torch.zeros(max(counts), rows, reg.p, dtype=torch.float64, device=eng.device)
    if mine:
        local[: len(mine)] = torch.stack([s.to(eng.device) for s in mine])
    if size == 1:
        return local[: counts[0]], counts
    gathered = [torch.empty_like(local) for _ in range(size)]
    dist.all_gather(gathered, local, group=group)
    return torch.cat([gathered[r][: counts[r]] for r in range(size)], dim=0), counts


def sharded_sample(reg, x, w=None, num_samples=100, latent=False, group=None):
    """Posterior samples split over ranks (`_sharded_sample_stack`).  Returns the full list of `num_samples` arrays on every
    rank (the reference's return type: host arrays, regression.py:564)."""
    stack, _ = _sharded_sample_stack(reg, x, w, num_samples, latent, group)
    host = stack.cpu().numpy()
    return [host[k] for k in range(host.shape[0])]


def sharded_predict(reg, x, w=None, num_samples=100, latent=False, credible_bounds=False, marginal=False, group=None):
    """`GPARRegressor.predict` with the samples drawn across the ranks and the Monte-Carlo reduction (mean, central 95 %
    bounds; reference regression.py:589-595) done on the device by `gpar_sample_stats` over the gathered stack - only the
    n* x p results cross to the host.  Every rank returns the same arrays."""
    stack, _ = _sharded_sample_stack(reg, x, w, num_samples, latent, group, marginal=marginal)
    eng = get_engine()
    if num_samples == 1:
        # the reference hands np.mean a single (n*, p) array here, so axis 0 is the input axis (regression.py:589-595); kept
        one = stack[0].cpu().numpy()
        mean = np.mean(one, axis=0)
        return (mean, np.percentile(one, 2.5, axis=0), np.percentile(one, 100 - 2.5, axis=0)) if credible_bounds else mean
    if credible_bounds:
        mean, lower, upper = eng.sample_stats(stack.contiguous(), 2.5, 100 - 2.5)
        return mean.cpu().numpy(), lower.cpu().numpy(), upper.cpu().numpy()
    return eng.sample_stats(stack.contiguous())[0].cpu().numpy()
