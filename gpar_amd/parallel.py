"""Layer-parallel GPAR over the GPUs of one node: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on the MI355X box, "gloo" in the CPU tests).

The p autoregressive layers are the units of work: layer i is owned by rank i mod G.  What has to travel between
ranks is decided by the model, exactly as in /root/reference/gpar/model.py:291-322:

  * `replace=False`, nothing to impute, no inducing points: layer i's design matrix is `[x, observed y_<i]`, which
    every rank already holds -> layers are independent, the data path has NO collective; only the p scalar
    log-likelihoods are summed (one 8-byte all-reduce).
  * otherwise (posterior means are fed forward): the owner of layer i computes the new input column (and the new
    inducing-input column) and broadcasts it — n x 8 bytes per layer, latency-bound — before layer i+1 can start.

`fit(fix=True)` trains each layer on its owner (the optimiser only touches names "{i}/*", reference
regression.py:453-454) and then broadcasts the trained latent variables so every rank holds the same `Vars`.
`predict` conditions every layer on every rank and splits the Monte-Carlo samples across ranks.
"""
import numpy as np
import torch
import torch.distributed as dist

from .engine import get_engine
from .model import host_masks, last, per_output

__all__ = ["world", "sharded_logpdf", "sharded_fit", "sharded_condition", "sharded_sample"]


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
    the collective (what makes a multi-GPU run interpretable: the slowest rank's busy time bounds the step)."""
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
    if timing is not None:
        timing["busy_s"] = timing.get("busy_s", 0.0) + (time.perf_counter() - t_start)
    if size > 1:
        buf = local.detach().to(device=eng.device, dtype=torch.float64).reshape(1).clone()
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        local = buf[0].cpu()
    return local


def _sharded_layers(gpar, x, y, w, x_ind, rank, size, group, local):
    from .model import _differentiable, _joining, _lockstep_values

    items = list(per_output(y, w, keep=gpar.impute))
    eng = get_engine()
    # this rank's layers alternate over two streams when no layer feeds another (see HipEngine.pipeline) ...
    pipe = eng.pipeline(rows=int(x.shape[0])) if gpar._independent(items) else None
    # ... or are factored together in lock-step
    lockstep = pipe is not None and gpar._same_rows(items) and hasattr(eng, "logpdf_dense_batch") and 0 < int(x.shape[0]) <= eng.batch_rows()
    pending = []
    values, stage = [], 0
    with _joining(pipe):
        for i, (is_last, ((yi, wi, mask), model)) in enumerate(last(zip(items, gpar.layers))):
            complete = isinstance(mask, slice)
            x = x[mask]
            mine = (i % size) == rank
            f = obs = None
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
                dist.broadcast(col, src=_global_rank(i % size, group), group=group)
                if ind_col is not None:
                    dist.broadcast(ind_col, src=_global_rank(i % size, group), group=group)
            x = torch.cat([x, col], dim=1)
            if ind_col is not None:
                x_ind = torch.cat([x_ind, ind_col], dim=1)
    if pipe is not None:
        pipe.join()
    if pending:
        values.extend(_lockstep_values(eng, pending))
    for v in values:
        local = local + v
    return local, x, x_ind


def _global_rank(group_rank, group):
    if group is None:
        return group_rank
    return dist.get_global_rank(group, group_rank)


def layers_train_independently(reg, y):
    """With `fix=True`, is the training of layer pi independent of the training of the other layers?  Its inputs must be
    data only (no `replace`, no inducing points, nothing to impute) and it must not share a hyper-parameter with
    another layer (`scale_tie` reads "0/input/scales", which only layer 0 trains)."""
    return not (reg.replace or reg.sparse or reg.model_config.get("scale_tie", False)
                or (reg.impute and bool(torch.isnan(y).any())))


def sharded_fit(reg, x, y, w=None, group=None, **kw_args):
    """`GPARRegressor.fit(x, y, w, fix=True)` with layer pi trained on rank pi mod G, then synchronised."""
    from .optimise import minimise_l_bfgs_b
    from .regression import _construct_gpar

    rank, size = world(group)
    eng = get_engine()
    reg.condition(x, y, w)
    x_dev, y_dev, w_dev = eng.tensor(reg.x), eng.tensor(reg.y), eng.tensor(reg.w)
    if isinstance(reg.y, np.ndarray) and y_dev.is_cuda and host_masks():
        y_dev._host_nan = np.isnan(reg.y)   # (masks planned on the host, see GPAR._prep)
    y_cached = {k: list(per_output(y_dev, w_dev, keep=k)) for k in [True, False]}
    if not layers_train_independently(reg, y_dev):
        # inputs of layer pi depend on the trained layers < pi: the chain is sequential; train replicated
        reg.fit(x, y, w, fix=True, **kw_args)
        return
    # instantiate every variable on every rank (lazy creation, reference regression.py:92-180)
    with torch.no_grad():
        _construct_gpar(reg, reg.vs, reg.m, reg.p).logpdf(x_dev[:2], y_dev[:2], w_dev[:2])
    for pi in range(reg.p):
        if pi % size != rank:
            continue
        gpar = _construct_gpar(reg, reg.vs, reg.m, pi + 1)
        fixed_x, fixed_x_ind = gpar.logpdf(x_dev, y_cached, None, only_last_layer=True, outputs=list(range(pi)), return_inputs=True)

        def objective(vs, pi=pi, fixed_x=fixed_x, fixed_x_ind=fixed_x_ind):
            g = _construct_gpar(reg, vs, reg.m, pi + 1)
            return -g.logpdf(fixed_x, y_cached, None, only_last_layer=True, outputs=[pi], x_ind=fixed_x_ind)

        minimise_l_bfgs_b(objective, reg.vs, names=[f"{pi}/*"], **kw_args)
    if size > 1:
        for pi in range(reg.p):
            names = reg.vs.match([f"{pi}/*"])
            if not names:
                continue
            vec = torch.as_tensor(reg.vs.get_vector(names), dtype=torch.float64).to(eng.device)
            dist.broadcast(vec, src=_global_rank(pi % size, group), group=group)
            reg.vs.set_vector(vec.cpu().numpy(), names)


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
    if size == 1 or not gpar._independent(items):
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
            # ragged layers (cannot happen in the independent regime this is used for): one broadcast per layer
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


def sharded_sample(reg, x, w=None, num_samples=100, latent=False, group=None):
    """Posterior samples split over ranks: the conditioning is layer-parallel (`sharded_condition`), each rank then draws
    its share with its own Philox stream, and the shares are all-gathered.  Returns the full list of `num_samples` arrays
    on every rank."""
    rank, size = world(group)
    eng = get_engine()
    counts = [num_samples // size + (1 if r < num_samples % size else 0) for r in range(size)]
    eng.seed(getattr(eng, "_seed", 0) * 1000003 + rank + 1)
    post = sharded_condition(reg, group) if size > 1 else None
    mine = reg.sample(x, w, posterior=True, num_samples=max(counts[rank], 1), latent=latent, _conditioned=post)
    mine = [mine] if isinstance(mine, np.ndarray) else list(mine)
    mine = mine[: counts[rank]]
    if size == 1:
        return mine
    rows = int(np.shape(x)[0])
    local = torch.zeros(max(counts), rows, reg.p, dtype=torch.float64, device=eng.device)
    for k, s in enumerate(mine):
        local[k] = torch.as_tensor(s, dtype=torch.float64)
    gathered = [torch.empty_like(local) for _ in range(size)]
    dist.all_gather(gathered, local, group=group)
    out = []
    for r in range(size):
        out.extend(gathered[r][k].cpu().numpy() for k in range(counts[r]))
    return out
