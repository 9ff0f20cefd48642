"""GPARRegressor: sklearn-style front end (drop-in for /root/reference/gpar/regression.py).

Same constructor keywords and defaults (reference regression.py:264-286), attributes (:288-326) and methods —
`get_variables`, `condition`, `fit`, `logpdf`, `sample`, `predict` — with the same argument meaning, return
types and exceptions.  What differs is underneath: the per-layer kernels are `gpar_amd.kernels` objects lowered
to one fused device kernel, and every Gram / Cholesky / solve runs in libgpar_hip.so on the MI355X.

Hyper-parameter names, initialisations and bounds follow reference regression.py:92-180 exactly (table in
SURVEY.md Appendix C), so `get_variables()` dictionaries are interchangeable.
"""
import os

import numpy as np
import torch

from .engine import get_engine
from .gp import GP, Measure
from .kernels import EQ, RQ, Linear, ZeroKernel
from .model import GPAR, host_masks, per_output
from .optimise import minimise_l_bfgs_b
from .vars import Vars

__all__ = ["GPARRegressor", "log_transform", "squishing_transform"]


def _xp(x):
    return torch if isinstance(x, torch.Tensor) else np


#: Log transform for the data: (transform, inverse).
log_transform = (lambda x: _xp(x).log(x), lambda x: _xp(x).exp(x))

#: Squishing transform for the data: sign(x) log(1 + |x|) and its inverse.
squishing_transform = (
    lambda x: _xp(x).sign(x) * _xp(x).log(1 + _xp(x).abs(x)),
    lambda x: _xp(x).sign(x) * (_xp(x).exp(_xp(x).abs(x)) - 1),
)


def _vector_from_init(init, length):
    """Broadcast a scalar initialisation, or take the first `length` entries of a vector one
    (reference regression.py:31-46; known answers tests/test_regression.py:43-49)."""
    if np.size(init) == 1:
        return init * np.ones(length)
    squeezed = np.squeeze(init)
    if np.ndim(squeezed) != 1:
        raise ValueError(f"Incorrect shape {np.shape(init)} of hyperparameters.")
    if np.size(squeezed) < length:
        raise ValueError("Not enough hyperparameters specified.")
    return np.array(squeezed)[:length]


def _determine_indices(m, pi, markov):
    """Columns of the design matrix used by layer `pi`: the m inputs and the last `markov` previous outputs
    (all of them for markov=None).  (reference regression.py:49-59; table tests/test_regression.py:52-83)"""
    p_last = pi - 1
    p_start = 0 if markov is None else max(p_last - (markov - 1), 0)
    p_num = p_last - p_start + 1
    return list(range(m)), list(range(m + p_start, m + p_last + 1)), p_num


def _to_torch(x):
    if x is None or isinstance(x, torch.Tensor):
        return x
    return torch.tensor(np.asarray(x))


def _to_engine(x):
    """A transient input (not kept as an attribute) goes to the engine's device before any arithmetic: element-wise host
    operators on n x p arrays open OpenMP regions whose workers then spin (see _default_weights)."""
    from .engine import get_engine

    return x if x is None else get_engine().tensor(x if isinstance(x, torch.Tensor) else np.asarray(x))


def _uprank(x):
    if x.dim() == 0:
        return x.reshape(1, 1)
    if x.dim() == 1:
        return x[:, None]
    return x


def _model_generator(vs, m, pi, scale, scale_tie, per, per_period, per_scale, per_decay, input_linear,
                     input_linear_scale, linear, linear_scale, nonlinear, nonlinear_scale, rq, markov, noise):
    """Constructor of layer `pi`: kernel over the inputs + kernel over the selected previous outputs, and the
    observation-noise variance; hyper-parameters are created in `vs` on first use (reference regression.py:72-182)."""

    config = repr((m, pi, scale, scale_tie, per, per_period, per_scale, per_decay, input_linear, input_linear_scale, linear,
                   linear_scale, nonlinear, nonlinear_scale, rq, markov, noise))

    def model():
        return vs.memo(config, build)

    def build():
        m_inds, p_inds, p_num = _determine_indices(m, pi, markov)
        k_in, k_out = ZeroKernel(), ZeroKernel()

        def nonlinear_kernel(prefix):
            return RQ(vs.bnd(name=f"{prefix}/alpha", init=1e-2, lower=1e-3, upper=1e3)) if rq else EQ()

        # nonlinear kernel over the inputs
        var = vs.bnd(name=f"{pi}/input/var", init=1.0)
        scales = vs.bnd(name=f"{0 if scale_tie else pi}/input/scales", init=_vector_from_init(scale, m))
        k_in = k_in + var * nonlinear_kernel(f"{pi}/input").stretch(scales)

        # locally periodic kernel over the inputs
        if per:
            var = vs.bnd(name=f"{pi}/input/per/var", init=1.0)
            scales = vs.bnd(name=f"{pi}/input/per/scales", init=_vector_from_init(per_scale, 2 * m))
            periods = vs.bnd(name=f"{pi}/input/per/pers", init=_vector_from_init(per_period, m))
            decays = vs.bnd(name=f"{pi}/input/per/decay", init=_vector_from_init(per_decay, m))
            k_in = k_in + var * EQ().stretch(scales).periodic(periods) * EQ().stretch(decays)

        # linear kernel (plus constant) over the inputs
        if input_linear:
            scales = vs.bnd(name=f"{pi}/input/lin/scales", init=_vector_from_init(input_linear_scale, m))
            const = vs.get(name=f"{pi}/input/lin/const", init=1.0)
            k_in = k_in + (Linear().stretch(scales) + const)

        # linear dependencies on the previous outputs
        if linear and pi > 0:
            scales = vs.bnd(name=f"{pi}/output/lin/scales", init=_vector_from_init(linear_scale, p_num))
            k_out = k_out + Linear().stretch(scales)

        # nonlinear dependencies on the previous outputs
        if nonlinear and pi > 0:
            var = vs.bnd(name=f"{pi}/output/nonlin/var", init=1.0)
            scales = vs.bnd(name=f"{pi}/output/nonlin/scales", init=_vector_from_init(nonlinear_scale, p_num))
            k_out = k_out + var * nonlinear_kernel(f"{pi}/output/nonlin").stretch(scales)

        noise_variance = vs.bnd(name=f"{pi}/noise", init=_vector_from_init(noise, pi + 1)[pi], lower=1e-8)
        f = GP(k_in.select(m_inds) + k_out.select(p_inds), measure=Measure())
        return f, noise_variance

    return model


def _construct_gpar(reg, vs, m, p):
    x_ind = reg.x_ind
    if x_ind is not None and getattr(reg, "_x_ind_trainable", False):
        # inducing inputs as a variable of the store (an addition: the reference keeps them fixed, todo.tasks:5)
        x_ind = vs.get(init=np.asarray(x_ind, dtype=np.float64), name="x_ind")
    gpar = GPAR(replace=reg.replace, impute=reg.impute, x_ind=x_ind, sparse_method=reg.sparse_method)
    for pi in range(p):
        gpar = gpar.add_layer(_model_generator(vs, m, pi, **reg.model_config))
    return gpar


def _default_weights(rows, cols):
    """The reference's default, a matrix of ones - made where it is used.  (A CPU `torch.ones` of 32768 elements or more
    opens an OpenMP parallel region; its worker threads - one per host core - then spin, and in a container with a CPU
    quota that throttles the process for tens of milliseconds: seen as one 60 ms evaluation in three at C4.)"""
    from .engine import get_engine

    return torch.ones(rows, cols, dtype=torch.float64, device=get_engine().device)


def _init_weights(w, y, attribute=False):
    if w is None:
        # (`condition` keeps the reference's attribute: a host tensor, made once)
        return torch.from_numpy(np.ones(tuple(y.shape), dtype=np.float64)) if attribute else _default_weights(*y.shape)
    return _uprank(_to_torch(w))


def _run_on_streams(eng, streams, shares, fn, on_exit=None):
    """fn(item) for every item of shares[k] on stream k, one host thread per stream; exceptions re-raised here.  `on_exit(k)`: called
    by thread k when it is done with its share, whatever happened (a lane of a lock-step rendezvous leaves it there)."""
    import threading

    main = torch.cuda.current_stream(eng.device)
    errors = []

    def work(k, stream, items):
        try:
            with torch.cuda.device(eng.device), torch.cuda.stream(stream):
                try:
                    for item in items:
                        fn(item)
                finally:
                    if on_exit is not None:
                        on_exit(k)
        except BaseException as exc:  # noqa: BLE001 - handed to the caller's thread
            errors.append(exc)

    threads = []
    for k, (stream, items) in enumerate(zip(streams, shares)):
        stream.wait_stream(main)
        threads.append(threading.Thread(target=work, args=(k, stream, items), daemon=True))
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for stream in streams:
        main.wait_stream(stream)
    if errors:
        raise errors[0]


class GPARRegressor:
    """GPAR regressor.  See the reference docstring (regression.py:200-262) for the meaning of every keyword;
    signature and defaults are identical."""

    #: `fit(fix=True)` trains dense layers through the prepared objective of gpar_amd/fastfit.py (same device launches, no
    #: autograd, no per-evaluation model objects); False keeps the general route for every layer (tests compare the two).
    fast_fit = True

    #: `logpdf` on device-resident outputs remembers their NaN pattern ON THE CALLER'S TENSOR (attribute `_gpar_nan`, with the tensor's
    #: version counter) so that a loop over the same outputs pays the device-to-host synchronisation once.  Writes torch cannot see
    #: - a raw-pointer kernel, another library, `.data` writes, a DLPack alias - do not move the counter: after one, `del y._gpar_nan`,
    #: or set this attribute False (per regressor, or on the class) and the pattern is read from the device at every call.
    nan_pattern_cache = True

    def __init__(self, replace=False, impute=True, scale=1.0, scale_tie=False, per=False, per_period=1.0,
                 per_scale=1.0, per_decay=10.0, input_linear=False, input_linear_scale=100.0, linear=True,
                 linear_scale=100.0, nonlinear=False, nonlinear_scale=1.0, rq=False, markov=None, noise=0.1,
                 x_ind=None, normalise_y=True, transform_y=(lambda x: x, lambda x: x), sparse_method="vfe"):
        self.replace = replace
        if sparse_method not in ("vfe", "fitc", "dtc"):
            raise ValueError('sparse_method must be "vfe", "fitc" or "dtc"')
        self.sparse_method = sparse_method  # an addition behind the reference's keywords: "vfe" | "fitc" | "dtc"
        self.impute = impute
        self.sparse = x_ind is not None
        self.x_ind = None if x_ind is None else _uprank(_to_torch(x_ind))
        self.model_config = {
            "scale": scale, "scale_tie": scale_tie, "per": per, "per_period": per_period, "per_scale": per_scale,
            "per_decay": per_decay, "input_linear": input_linear, "input_linear_scale": input_linear_scale,
            "linear": linear, "linear_scale": linear_scale, "nonlinear": nonlinear,
            "nonlinear_scale": nonlinear_scale, "rq": rq, "markov": markov, "noise": noise,
        }
        self.vs = Vars(dtype=torch.float64)
        self.is_conditioned = False
        self.x = self.y = self.w = None
        self.n = self.m = self.p = None
        self.normalise_y = normalise_y
        self._unnormalise_y, self._normalise_y = (lambda x: x), (lambda x: x)
        self._transform_y, self._untransform_y = transform_y

    def get_variables(self):
        """Dictionary name -> value of every hyper-parameter instantiated so far."""
        return {name: self.vs[name].detach().numpy() for name in self.vs.names}

    def condition(self, x, y, w=None):
        """Store (and transform / normalise) the training data without training (reference regression.py:339-389)."""
        # The attributes are host tensors, as in the reference.  The element-wise passes over them (mask, mean, standard
        # deviation, normalisation) go through numpy, which runs them on the calling thread: a torch CPU operator on 32768 or
        # more elements opens an OpenMP region whose workers (one per core) spin afterwards, which in a container with a CPU
        # quota throttles the evaluations that follow (fit(iters=3) at n = 8192: 0.68 -> 0.60 s) - and changing torch's
        # global thread count around the call instead would be visible to other threads of the process.
        self.x = _uprank(_to_torch(x)).detach().cpu()
        self.y = self._transform_y(_uprank(_to_torch(y))).detach().cpu()
        self.w = _init_weights(w, self.y, attribute=True)
        self.n, self.m = self.x.shape
        self.p = self.y.shape[1]
        if self.normalise_y:
            y_np = self.y.detach().cpu().numpy()   # (a device tensor handed to fit / condition comes to the host here)
            means = np.empty((1, self.p), dtype=y_np.dtype)
            stds = np.empty((1, self.p), dtype=y_np.dtype)
            for i in range(self.p):
                y_i = y_np[~np.isnan(y_np[:, i]), i]
                means[0, i] = np.mean(y_i)
                std = np.std(y_i)  # population std, as lab's B.std
                stds[0, i] = std if std > 0 else 1.0
            means_t, stds_t = torch.from_numpy(means), torch.from_numpy(stds)

            def normalise_y(y_):
                return (y_ - means_t.to(y_.device)) / stds_t.to(y_.device)

            def unnormalise_y(y_):
                return y_ * stds_t.to(y_.device) + means_t.to(y_.device)

            self._normalise_y, self._unnormalise_y = normalise_y, unnormalise_y
            self.y = torch.from_numpy((y_np - means) / stds)
        self.is_conditioned = True

    def fit(self, x, y, w=None, greedy=False, fix=True, optimise_x_ind=False, **kw_args):
        """Train layer by layer with L-BFGS-B on the negative log marginal likelihood; keyword arguments go to
        `minimise_l_bfgs_b` (`iters`, `f_calls`, `trace`).  (reference regression.py:391-459)

        `optimise_x_ind` (an addition, off by default; the reference's todo.tasks:5): the inducing inputs `x_ind` become the
        variable "x_ind" of `self.vs` and are trained along with every layer's hyper-parameters (the gradient with respect to
        inducing locations comes from the same device passes as the joint gradient of `fix=False`)."""
        self.condition(x, y, w)
        if greedy:
            # (as the reference, regression.py:409-410 and its test tests/test_regression.py:241-243; the search itself is
            # `greedy_order` below - an addition, not a change of this call)
            raise NotImplementedError("Greedy search is not implemented yet.")
        if optimise_x_ind and not self.sparse:
            raise ValueError("optimise_x_ind needs inducing points (x_ind)")
        self._train(range(self.p), fix, optimise_x_ind, **kw_args)

    def _train(self, layers, fix=True, optimise_x_ind=False, concurrent=True, **kw_args):
        """Train the given layers of the conditioned model, one after the other (or, where they do not feed one another, on the
        engine's worker streams); returns {layer: final value of its objective}."""
        self._x_ind_trainable = bool(optimise_x_ind) or getattr(self, "_x_ind_trainable", False)
        layers = list(layers)
        finals = {}
        eng = get_engine()
        x_dev, y_dev, w_dev = eng.tensor(self.x), eng.tensor(self.y), eng.tensor(self.w)
        if host_masks():
            # self.y is a host tensor (condition): its NaN pattern costs nothing here, and per_output then plans the masks on the
            # host - index tensors, no synchronisation per layer and evaluation (on any engine: the CPU tests walk the same route)
            if y_dev is self.y:
                y_dev = y_dev.view(y_dev.shape)   # (never hang the plan on the regressor's own attribute)
            y_dev._host_nan, y_dev._host_nan_version = torch.isnan(self.y).numpy(), y_dev._version
        y_cached = {k: list(per_output(y_dev, w_dev, keep=k)) for k in [True, False]}
        self._prepare_kernels(self.m, self.p, self.n, training=True, inputs=not fix or bool(optimise_x_ind))

        def train_layer(pi, group=None, lane=None):
            if fix:
                gpar = _construct_gpar(self, self.vs, self.m, pi + 1)
                fixed_x, fixed_x_ind = gpar.logpdf(
                    x_dev, y_cached, None, only_last_layer=True, outputs=list(range(pi)), return_inputs=True
                )

            def objective(vs):
                gpar = _construct_gpar(self, vs, self.m, pi + 1)
                if fix:
                    x_ind_pi = fixed_x_ind
                    if optimise_x_ind:  # the m base columns are the variable; the columns appended by earlier layers stay fixed
                        x_ind_pi = torch.cat([eng.tensor(vs["x_ind"]), fixed_x_ind[:, self.m :]], dim=1)
                    return -gpar.logpdf(fixed_x, y_cached, None, only_last_layer=True, outputs=[pi], x_ind=x_ind_pi)
                return -gpar.logpdf(x_dev, y_cached, None, only_last_layer=False)

            names = [f"{pi}/*"] if fix else [f"{i}/*" for i in range(pi + 1)]
            if optimise_x_ind:
                names = names + ["x_ind"]
            fast = None
            if fix and not optimise_x_ind and self.fast_fit:
                # a dense layer with fixed inputs: the objective prepared once, an evaluation = one library call + the chain rule
                # in numpy (gpar_amd/fastfit.py); None where that route does not apply
                from . import fastfit
                from .optimise import objective_and_gradient

                general = []

                def general_fg(x):   # (a failed factorisation goes through the general route's evaluation: unfused retry, NaN)
                    if not general:
                        general.append(objective_and_gradient(objective, self.vs, names, trace=kw_args.get("trace", False))[0])
                    return general[0](x)

                fast = fastfit.build(self, eng, self.vs, pi, names, fixed_x, y_cached[bool(self.impute)][pi], general_fg=general_fg,
                                     group=group, lane=lane)
            if group is not None and (fast is None or fast.group is None):
                group.leave(lane)   # (this lane trains through the general route from here on: the others must not wait for it)
            if fast is not None:
                finals[pi] = fast.minimise(**kw_args)
            else:
                finals[pi] = minimise_l_bfgs_b(objective, self.vs, names=names, **kw_args)
            if optimise_x_ind and "x_ind" in self.vs:
                self.x_ind = self.vs["x_ind"].detach().clone()

        from .parallel import layers_train_independently

        depth, prepared = None, False
        if fix and not optimise_x_ind and self.fast_fit and not self.sparse:
            from .gp import one_call_grad_rows

            prepared = 0 < self.n <= one_call_grad_rows()   # every layer goes through the prepared objective (fastfit.py)
            if prepared and os.environ.get("GPAR_FIT_THREADS") is None:
                # the prepared objective leaves ~0.1 ms of interpreter time per evaluation: four drivers no longer contend for it
                # (fit(iters=20), four layers, 2 -> 4 threads: n = 100 21 -> 15 ms, 400 35 -> 29, 1024 43 -> 30; profiles/r06_small_fit.txt)
                depth = 4
        streams = eng.worker_streams(depth=depth, rows=self.n) if hasattr(eng, "worker_streams") else []
        if concurrent and fix and len(layers) > 1 and len(streams) > 1 and layers_train_independently(self, y_dev):
            # Layers whose inputs are data and whose hyper-parameters are their own train independently of one another:
            # two host threads, each on its own stream, keep two L-BFGS-B drivers in flight so that one layer's
            # latency-bound stretches (panel chains, host-side optimiser steps) run under the other's GEMMs.  The result
            # is the serial one: every evaluation is the same deterministic device computation.
            with torch.no_grad():  # lazily created variables must all exist before the store is shared between threads
                _construct_gpar(self, self.vs, self.m, self.p).logpdf(x_dev[:2], y_dev[:2], w_dev[:2])
            lanes = min(len(streams), len(layers))
            shares = [layers[k::lanes] for k in range(lanes)]
            group = None
            if prepared and lanes > 1 and all(isinstance(item[2], slice) for item in y_cached[bool(self.impute)]):
                # every layer goes through the prepared objective on the same number of rows: from ~1000 rows on their
                # factorisations are taken in lock-step, one gpar_potrf_batch per round of evaluations (fastfit.LockstepFactor)
                from . import fastfit

                lo, hi = fastfit.lockstep_rows()
                if lo <= self.n <= hi and hasattr(eng, "_grads_from_moments"):
                    group = fastfit.LockstepFactor(eng, self.n, lanes)
            if group is None:
                _run_on_streams(eng, streams[:lanes], shares, train_layer)
            else:
                tagged = [[(pi, k) for pi in share] for k, share in enumerate(shares)]
                _run_on_streams(eng, streams[:lanes], tagged, lambda item: train_layer(item[0], group, item[1]), on_exit=group.leave)
                self._lockstep_rounds = (group.rounds, group.batches)
        else:
            for pi in layers:
                train_layer(pi)
        return finals

    def greedy_order(self, x, y, w=None, **kw_args):
        """Greedy search for the ORDER of the outputs - the reference's open item (regression.py:400 `greedy`, :409-410
        NotImplementedError, todo.tasks:8; Requeima et al. 2019, section 5: "greedily select the output that maximises the
        log marginal likelihood conditioned on the already selected ones") as an ADDITION: `fit(greedy=True)` keeps raising.

        Position by position: every remaining output is trained as the next layer - inputs x and the outputs chosen so far,
        exactly the layer `fit(fix=True)` would train there, same initial values, same L-BFGS-B keyword arguments (`iters`,
        `f_calls`, `trace`) - and the one with the largest trained log marginal likelihood of that layer is kept.  That is
        p (p + 1) / 2 layer fits; the candidates of a position are independent of one another and are trained concurrently on the
        engine's worker streams where layers do not feed one another (complete data, no `replace`, no inducing points).

        Returns (order, values): `order[i]` is the column of y placed at position i, `values[i]` the log marginal likelihood of
        that layer after training (of the normalised outputs, as `fit` sees them).  The trained hyper-parameters of the chosen
        chain are kept in `self.greedy_vs_` (layer i there belongs to output `order[i]`): `reg.vs = reg.greedy_vs_.copy();
        reg.fit(x, y[:, order], ...)` continues from them.  `self` is conditioned on (x, y, w) in the GIVEN order, as after
        `condition`."""
        from .vars import Vars

        self.condition(x, y, w)
        eng = get_engine()
        x_host, y_host, w_host = self.x, self.y, self.w
        order, values, remaining = [], [], list(range(self.p))
        chain_vs = Vars(dtype=torch.float64)
        config = dict(self.model_config)
        streams = eng.worker_streams(rows=self.n) if hasattr(eng, "worker_streams") else []
        independent = not self.replace and not self.sparse and not bool(torch.isnan(y_host).any())
        for k in range(self.p):
            results = {}

            def trial(c):
                cols = order + [c]
                reg = GPARRegressor(replace=self.replace, impute=self.impute, x_ind=self.x_ind, normalise_y=False,
                                    sparse_method=self.sparse_method, **config)
                reg.vs = chain_vs.copy(detach=True)
                reg.condition(x_host, y_host[:, cols], w_host[:, cols])
                final = reg._train([k], fix=True, concurrent=False, **kw_args)[k]
                results[c] = (-float(final), reg.vs)

            if independent and len(streams) > 1 and len(remaining) > 1:
                lanes = min(len(streams), len(remaining))
                _run_on_streams(eng, streams[:lanes], [remaining[j::lanes] for j in range(lanes)], trial)
            else:
                for c in remaining:
                    trial(c)
            # (ties - identical columns - go to the lower column index: the order of `remaining`; a trial whose optimisation ended on
            # a failed evaluation carries NaN, which would make the comparisons order-dependent: it ranks below every finite value)
            def rank(c):
                value = results[c][0]
                return (value if np.isfinite(value) else -np.inf, -c)

            if not any(np.isfinite(results[c][0]) for c in remaining):
                raise ArithmeticError(f"greedy_order: every candidate for position {k} ended on a failed evaluation")
            best = max(remaining, key=rank)
            order.append(best)
            values.append(results[best][0])
            chain_vs = results[best][1]
            remaining.remove(best)
        self.greedy_vs_ = chain_vs
        self.greedy_order_ = list(order)
        return order, values

    def _prepare_kernels(self, m, p, rows, training=False, inputs=False):
        """Have the engine compile the layers' run-time specialised device kernels up front and concurrently (HipEngine.prepare);
        the layer constructors instantiate their hyper-parameters on the way, as the first evaluation would."""
        eng = get_engine()
        cols = int(self.x_ind.shape[0]) if self.sparse else rows   # (inducing points: the launches are rows x M)
        if not hasattr(eng, "prepare") or rows * cols < (1 << 22) or os.environ.get("GPAR_JIT_PREPARE", "1") == "0":   # (prepare applies the thresholds)
            return
        with torch.no_grad():
            layers = _construct_gpar(self, self.vs, m, p).layers
            eng.prepare([(model()[0].kernel, m + pi) for pi, model in enumerate(layers)], rows, training=training, sparse=self.sparse,
                        inputs=inputs, cols=cols)

    def logpdf(self, x, y, w=None, sample_missing=False, posterior=False):
        """Log-density of observations under the prior (or, with `posterior`, the conditioned model).  Returns a
        numpy scalar unless x or y was a torch tensor (reference regression.py:461-506)."""
        any_torch = isinstance(x, torch.Tensor) or isinstance(y, torch.Tensor)
        y_given = y
        x = _uprank(_to_engine(x))
        y = self._unnormalise_y(self._transform_y(_uprank(_to_engine(y))))  # sic: reference regression.py:483
        if isinstance(y_given, torch.Tensor) and y_given.is_cuda and y.dim() == 2 and y.data_ptr() == y_given.data_ptr() and host_masks():
            # Device-resident outputs handed over as they are (no transform, no normalisation: `y` is an alias of the caller's
            # tensor): their NaN pattern decides every mask of the evaluation and costs a device-to-host synchronisation, which a
            # loop over the same outputs (an optimiser, a benchmark) would pay every time.  It is kept ON THE CALLER'S TENSOR OBJECT
            # together with the version counter it was taken at - it lives and dies with that object, nothing global holds the
            # tensor, and an in-place torch operation is seen.  (A write torch cannot see - a raw-pointer kernel, `.data`, a DLPack
            # alias - is not: `del y._gpar_nan` after one, or switch the cache off with `nan_pattern_cache = False`.)
            cached = getattr(y_given, "_gpar_nan", None) if self.nan_pattern_cache else None
            if cached is None or cached[0] != y_given._version or cached[1].shape != tuple(y.shape):
                cached = (y_given._version, torch.isnan(y).cpu().numpy())
                if self.nan_pattern_cache:
                    try:
                        y_given._gpar_nan = cached
                    except (AttributeError, RuntimeError):
                        pass
            y._host_nan, y._host_nan_version = cached[1], y._version
        w = _init_weights(w, y)
        m, p = x.shape[1], y.shape[1]
        if posterior and not self.is_conditioned:
            raise RuntimeError("Must condition or fit model before computing the logpdf under the posterior.")
        self._prepare_kernels(m, p, int(x.shape[0]))
        gpar = _construct_gpar(self, self.vs, m, p)
        if posterior:
            gpar = gpar | (self.x, self.y, self.w)
        value = gpar.logpdf(x, y, w, only_last_layer=False, sample_missing=sample_missing)
        if not any_torch:
            value = value.detach().cpu().numpy()
        return value

    def _sample_device(self, x, w, p, posterior, num_samples, latent, conditioned=None, marginal=False):
        """The samples of `sample` as engine tensors (n* x p each), output transforms undone."""
        x = _uprank(_to_engine(x))
        if posterior and not self.is_conditioned:
            raise RuntimeError("Must condition or fit model before sampling from the posterior.")
        elif not posterior and p is None:
            raise ValueError("Must specify number of outputs to sample.")
        if w is None:
            w = _default_weights(x.shape[0], self.p if posterior else p)
        else:
            w = _uprank(_to_torch(w))
        if posterior and conditioned is not None:
            gpar = conditioned  # parallel.sharded_condition: factors computed across ranks
        elif posterior:
            gpar = _construct_gpar(self, self.vs, self.m, self.p)
            gpar = gpar | (self.x, self.y, self.w)
        else:
            gpar = _construct_gpar(self, self.vs, x.shape[1], p)
        return [self._untransform_y(self._unnormalise_y(s)).detach()
                for s in gpar.sample_many(x, w, num_samples, latent=latent, marginal=marginal)]

    def predict_moments(self, x, w=None, latent=False, _conditioned=None):
        """Predictive means and variances (two n* x p arrays) of the conditioned model at x in CLOSED FORM - for `replace=True`,
        where posterior means are fed forward and the predictive law of every output at every point is Gaussian.  These are
        the limits of `predict`'s Monte-Carlo mean and of the variance of its samples (central 95 % bounds: mean -+ 1.96 sd);
        no sampling, no n* x n* covariance.  An addition behind the reference's API (it only samples, regression.py:566-597);
        raises ValueError for `replace=False`, where sampled values are fed forward and no closed form exists."""
        if not self.is_conditioned:
            raise RuntimeError("Must condition or fit model before predicting.")
        if not self.replace:
            raise ValueError("closed-form predictive moments need replace=True")
        probe = torch.tensor([-1.5, 0.25, 3.0], dtype=torch.float64)
        if not torch.equal(self._untransform_y(probe), probe):
            raise ValueError("closed-form predictive moments need the identity output transform (a non-linear map of a Gaussian is not Gaussian)")
        x = _uprank(_to_engine(x))
        w = _default_weights(x.shape[0], self.p) if w is None else _uprank(_to_torch(w))
        gpar = _conditioned
        if gpar is None:
            gpar = _construct_gpar(self, self.vs, self.m, self.p) | (self.x, self.y, self.w)
        with torch.no_grad():
            mean, var = gpar.moments(x, w, latent=latent)
            # un-normalisation is affine (y = y_n * std + mean): apply it to the mean, its slope squared to the variance
            zero = torch.zeros(1, self.p, dtype=mean.dtype, device=mean.device)
            one = torch.ones(1, self.p, dtype=mean.dtype, device=mean.device)
            shift = self._unnormalise_y(zero)
            slope = self._unnormalise_y(one) - shift
            mean = self._unnormalise_y(mean)
            var = var * slope ** 2
        return mean.cpu().numpy(), var.cpu().numpy()

    def sample(self, x, w=None, p=None, posterior=False, num_samples=1, latent=False, _conditioned=None):
        """Draw samples from the prior or the posterior at inputs x; a single ndarray for num_samples=1, otherwise
        a list (reference regression.py:508-564).  (`_conditioned`: an already conditioned GPAR, used by the
        multi-GPU path.)"""
        samples = [s.cpu().numpy() for s in self._sample_device(x, w, p, posterior, num_samples, latent, _conditioned)]
        return samples[0] if num_samples == 1 else samples

    def predict(self, x, w=None, num_samples=100, latent=False, credible_bounds=False, marginal=False):
        """Monte-Carlo predictive mean (and central 95% marginal bounds) from posterior samples
        (reference regression.py:566-597).  The reduction over the sample axis runs on the device
        (`gpar_sample_stats`: sequential mean, numpy-"linear" percentiles); only the n* x p results cross PCIe.

        `marginal=True` (an addition, off by default; reference todo.tasks:5): the statistics returned here only involve
        per-point marginals, so within each layer the points may be drawn from their marginals N(mean_j, var_j)
        instead of jointly - same distribution of every returned number, no n* x n* covariance, SYRK downdate or
        factorisation per sample (`GPAR.sample_many`)."""
        samples = self._sample_device(x, w, None, True, num_samples, latent, marginal=marginal)
        if num_samples == 1:
            # the reference hands np.mean a single (n*, p) array here, so axis 0 is the input axis; keep that
            samples = samples[0].cpu().numpy()
            mean = np.mean(samples, axis=0)
            if credible_bounds:
                return mean, np.percentile(samples, 2.5, axis=0), np.percentile(samples, 100 - 2.5, axis=0)
            return mean
        stack = torch.stack(samples)
        if credible_bounds:
            mean, lower, upper = get_engine().sample_stats(stack, 2.5, 100 - 2.5)
            return mean.cpu().numpy(), lower.cpu().numpy(), upper.cpu().numpy()
        return get_engine().sample_stats(stack)[0].cpu().numpy()
