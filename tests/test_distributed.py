"""N > 1 path on CPU: two processes, `gloo` backend, oracle engine.  Checks that the layer-sharded log marginal
likelihood equals the serial one in both regimes (independent layers: no data-path collective; dependent chain:
forwarded columns broadcast from the layer's owner), that layer-parallel training reproduces serial training, and
that sample-parallel prediction returns the requested number of samples on every rank."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _data(seed=0, n=18, p=3, missing=False):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, (n, 2))
    y = np.stack([np.sin(3 * x[:, 0]), np.cos(2 * x[:, 1]) + x[:, 0], x[:, 0] * x[:, 1]], axis=1)[:, :p]
    y = y + 0.05 * rng.standard_normal(y.shape)
    if missing:
        y[[2, 5], 0] = np.nan
        y[[7], 1] = np.nan
    w = rng.random((n, p)) + 0.5
    return x, y, w


def _worker(rank, size, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        from gpar_amd.engine import set_engine
        from gpar_amd.parallel import sharded_fit, sharded_logpdf, sharded_sample
        from gpar_amd.regression import GPARRegressor, _construct_gpar
        from oracle.engine import OracleEngine

        set_engine(OracleEngine(seed=5))
        out = {}
        for name, kw, missing in [
            ("independent", dict(replace=False, impute=False), False),
            ("impute-chain", dict(replace=False, impute=True), True),
            ("replace-chain", dict(replace=True, impute=True), False),
            ("sparse-chain", dict(replace=False, impute=False, x_ind=np.random.default_rng(1).uniform(-1, 1, (6, 2))), False),
        ]:
            x, y, w = _data(missing=missing)
            reg = GPARRegressor(nonlinear=True, noise=0.05, normalise_y=False, **kw)
            gpar = _construct_gpar(reg, reg.vs, 2, 3)
            serial = float(gpar.logpdf(x, y, w))
            sharded = float(sharded_logpdf(gpar, x, y, w))
            out[name] = (serial, sharded)
        # layer-parallel training == serial training (independent regime)
        x, y, w = _data()
        a = GPARRegressor(nonlinear=True, noise=0.1, impute=False)
        b = GPARRegressor(nonlinear=True, noise=0.1, impute=False)
        a.fit(x, y, w, iters=8)
        sharded_fit(b, x, y, w, iters=8)
        va, vb = a.get_variables(), b.get_variables()
        out["fit"] = (sorted(va) == sorted(vb), max(float(np.max(np.abs(va[k] - vb[k]))) for k in va))
        samples = sharded_sample(b, x[:5], None, num_samples=5)
        out["samples"] = (len(samples), samples[0].shape)
        # layer-parallel conditioning (factors computed by their owners, broadcast to everybody) == local conditioning
        from gpar_amd.engine import get_engine
        from gpar_amd.parallel import sharded_condition

        post = sharded_condition(b)
        get_engine().seed(99)
        via_exchange = np.stack(b.sample(x[:7], posterior=True, num_samples=3, _conditioned=post))
        get_engine().seed(99)
        local = np.stack(b.sample(x[:7], posterior=True, num_samples=3))
        received = [i for i, layer in enumerate(post.layers) if layer()[0]._obs._fac.logdet is None]
        out["condition"] = (float(np.max(np.abs(via_exchange - local))), received)
        results[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_layer_parallel_matches_serial_world_size_2():
    size = 2
    ctx = mp.get_context("spawn")
    manager = ctx.Manager()
    results = manager.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, size, port, results)) for r in range(size)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=570)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    for rank in range(size):
        out = results[rank]
        for name in ("independent", "impute-chain", "replace-chain", "sparse-chain"):
            serial, sharded = out[name]
            assert abs(serial - sharded) <= 1e-10 * abs(serial), (rank, name, serial, sharded)
        same_names, maxdiff = out["fit"]
        assert same_names and maxdiff < 1e-9
        assert out["samples"] == (5, (5, 3))
        maxdiff, received = out["condition"]
        assert maxdiff == 0.0, maxdiff
        assert received == [i for i in range(3) if i % size != rank], (rank, received)  # the others' layers came over the wire
    # both ranks hold the same totals
    assert results[0]["independent"][1] == results[1]["independent"][1]


def _worker_c3(rank, size, port, results):
    """BASELINE C3's partitioning at toy size: p = 8 layers, markov = 2, four ranks -> two layers per rank."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        from gpar_amd.engine import get_engine, set_engine
        from gpar_amd.parallel import sharded_condition, sharded_fit, sharded_logpdf, sharded_sample
        from gpar_amd.regression import GPARRegressor, _construct_gpar
        from oracle.engine import OracleEngine

        set_engine(OracleEngine(seed=3))
        rng = np.random.default_rng(4)
        n, m, p = 24, 4, 8
        x = rng.uniform(0, 1, (n, m))
        cols = []
        for i in range(p):
            f = np.sin(3 * x @ rng.uniform(0.5, 1.5, m) + i)
            if cols:
                f = f + 0.5 * np.cos(cols[-1])
            cols.append(f + 0.05 * rng.standard_normal(n))
        y = np.stack(cols, axis=1)
        kw = dict(scale=0.5, linear=True, nonlinear=True, markov=2, noise=0.1, normalise_y=False)
        reg = GPARRegressor(**kw)
        gpar = _construct_gpar(reg, reg.vs, m, p)
        out = {"logpdf": (float(gpar.logpdf(x, y, np.ones_like(y))), float(sharded_logpdf(gpar, x, y, np.ones_like(y))))}
        a, b = GPARRegressor(**kw), GPARRegressor(**kw)
        a.fit(x, y, iters=5)
        sharded_fit(b, x, y, iters=5)
        va, vb = a.get_variables(), b.get_variables()
        out["fit"] = (sorted(va) == sorted(vb), max(float(np.max(np.abs(va[k] - vb[k]))) for k in va))
        post = sharded_condition(b)
        received = [i for i, layer in enumerate(post.layers) if layer()[0]._obs._fac.logdet is None]
        get_engine().seed(7)
        via = np.stack(b.sample(x[:6], posterior=True, num_samples=2, _conditioned=post))
        get_engine().seed(7)
        local = np.stack(b.sample(x[:6], posterior=True, num_samples=2))
        out["condition"] = (float(np.max(np.abs(via - local))), received)
        out["samples"] = len(sharded_sample(b, x[:5], None, num_samples=6))
        results[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_c3_partitioning_world_size_4():
    """p = 8, markov = 2 on four ranks (layers i and i + 4 on rank i): log-likelihood, training, packed all-gather of the
    factors (two rounds of four layers) and sample-parallel prediction."""
    size = 4
    ctx = mp.get_context("spawn")
    manager = ctx.Manager()
    results = manager.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_c3, args=(r, size, port, results)) for r in range(size)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=570)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    for rank in range(size):
        out = results[rank]
        serial, sharded = out["logpdf"]
        assert abs(serial - sharded) <= 1e-10 * abs(serial), (rank, serial, sharded)
        same, maxdiff = out["fit"]
        assert same and maxdiff < 1e-9
        maxdiff, received = out["condition"]
        assert maxdiff == 0.0
        assert received == [i for i in range(8) if i % size != rank]
        assert out["samples"] == 6
    assert len({results[r]["logpdf"][1] for r in range(size)}) == 1


def _worker_random(rank, size, port, results):
    """Randomly drawn model options / missing patterns / weights / inducing points on an ODD number of ranks, p not a multiple of it."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        from gpar_amd.engine import set_engine
        from gpar_amd.parallel import sharded_logpdf
        from gpar_amd.regression import GPARRegressor, _construct_gpar
        from oracle.engine import OracleEngine

        set_engine(OracleEngine(seed=5))
        out = []
        for seed in range(14):
            rng = np.random.default_rng(300 + seed)   # (every rank draws the same case)
            n, m, p = int(rng.integers(3, 40)), int(rng.integers(1, 3)), int(rng.integers(1, 6))
            kw = dict(scale=float(rng.uniform(0.3, 1.2)), noise=float(rng.uniform(0.05, 0.4)), normalise_y=False, linear=bool(rng.integers(2)),
                      nonlinear=bool(rng.integers(2)), rq=bool(rng.integers(2)), markov=[None, 0, 1, 2][int(rng.integers(4))],
                      impute=bool(rng.integers(2)), replace=bool(rng.integers(3) == 0))
            if rng.integers(3) == 0 and n >= 6:
                kw["x_ind"] = rng.uniform(0, 1, (int(rng.integers(2, 6)), m))
            x = rng.uniform(0, 1, (n, m))
            y = np.stack([np.sin(3 * x[:, 0] + i) + 0.1 * rng.standard_normal(n) for i in range(p)], axis=1)
            if rng.integers(2):
                y[rng.random(y.shape) < 0.2] = np.nan
                y[0] = 0.1
            w = rng.uniform(0.5, 2.0, (n, p))
            reg = GPARRegressor(**kw)
            gpar = _construct_gpar(reg, reg.vs, m, p)
            out.append((float(gpar.logpdf(x, y, w)), float(sharded_logpdf(gpar, x, y, w))))
        results[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_random_configurations_on_three_ranks():
    size = 3
    ctx = mp.get_context("spawn")
    manager = ctx.Manager()
    results = manager.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_random, args=(r, size, port, results)) for r in range(size)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=570)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    for rank in range(size):
        for serial, sharded in results[rank]:
            assert abs(serial - sharded) <= 1e-10 * max(abs(serial), 1.0), (rank, serial, sharded)
    assert [s for _, s in results[0]] == [s for _, s in results[1]] == [s for _, s in results[2]]   # every rank holds the same totals


def _worker_joint(rank, size, port, results):
    """`fit(fix=False)` with the sum over layers divided over the ranks; Markov-limited forwarding; device-side predict."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        from gpar_amd.engine import get_engine, set_engine
        from gpar_amd.parallel import forward_plan, sharded_fit, sharded_logpdf, sharded_predict, sharded_sample
        from gpar_amd.regression import GPARRegressor, _construct_gpar
        from oracle.engine import OracleEngine

        set_engine(OracleEngine(seed=5))
        out = {}
        x, y, w = _data(n=16)
        for name, kw in [("joint", dict(nonlinear=True, noise=0.1, impute=False)),
                         ("joint-tied", dict(nonlinear=True, noise=0.1, impute=False, scale_tie=True, markov=1))]:
            a, b = GPARRegressor(**kw), GPARRegressor(**kw)
            a.fit(x, y, w, fix=False, iters=6)
            mode = sharded_fit(b, x, y, w, fix=False, iters=6)
            va, vb = a.get_variables(), b.get_variables()
            out[name] = (mode, sorted(va) == sorted(vb), max(float(np.max(np.abs(va[k] - vb[k]))) for k in va),
                         {k: vb[k].tolist() for k in vb})
        # dependent regimes train replicated, and say so
        c = GPARRegressor(nonlinear=True, noise=0.1, replace=True)
        out["replicated"] = (sharded_fit(c, x, y, w, fix=False, iters=2), sharded_fit(GPARRegressor(noise=0.1, scale_tie=True, impute=False), x, y, w, iters=2))
        # Markov-limited forwarding: p = 5, markov = 1, dependent chain (replace): column i goes to the owner of layer i + 1 only
        rng = np.random.default_rng(11)
        x5 = rng.uniform(-1, 1, (14, 2))
        y5 = np.stack([np.sin(2 * x5[:, 0] + i) + 0.1 * rng.standard_normal(14) for i in range(5)], axis=1)
        w5 = np.ones_like(y5)
        plans = {}
        for name, kw in [("markov1", dict(replace=True, markov=1)), ("markov2-sparse", dict(markov=2, x_ind=rng.uniform(-1, 1, (5, 2)))),
                         ("markov0", dict(replace=True, markov=0)), ("full", dict(replace=True))]:
            reg = GPARRegressor(nonlinear=True, noise=0.05, normalise_y=False, **kw)
            gpar = _construct_gpar(reg, reg.vs, 2, 5)
            plans[name] = [sorted(s) for s in forward_plan(gpar, 2, size)]
            out[name] = (float(gpar.logpdf(x5, y5, w5)), float(sharded_logpdf(gpar, x5, y5, w5)))
        out["plans"] = plans
        # predict: the device-side reduction over the gathered stack == numpy over the gathered samples
        get_engine().seed(21)
        mean, lo, hi = sharded_predict(b, x[:6], num_samples=7, credible_bounds=True)
        get_engine().seed(21)
        samples = np.stack(sharded_sample(b, x[:6], num_samples=7))
        out["predict"] = (float(np.max(np.abs(mean - samples.mean(axis=0)))), float(np.max(np.abs(lo - np.percentile(samples, 2.5, axis=0)))),
                          float(np.max(np.abs(hi - np.percentile(samples, 97.5, axis=0)))), mean.tolist())
        results[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("size", [2, 3])
def test_joint_fit_markov_sends_and_device_predict(size):
    ctx = mp.get_context("spawn")
    manager = ctx.Manager()
    results = manager.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker_joint, args=(r, size, port, results)) for r in range(size)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=850)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    for rank in range(size):
        out = results[rank]
        for name in ("joint", "joint-tied"):
            mode, same, maxdiff, _ = out[name]
            assert mode == "joint-sharded" and same and maxdiff < 1e-9, (rank, name, mode, maxdiff)
        assert out["replicated"] == ("replicated", "replicated")
        for name in ("markov1", "markov2-sparse", "markov0", "full"):
            serial, sharded = out[name]
            assert abs(serial - sharded) <= 1e-10 * abs(serial), (rank, name, serial, sharded)
        plans = out["plans"]
        assert plans["markov1"] == [[(i + 1) % size] for i in range(4)] + [[]]
        assert plans["markov0"] == [[] for _ in range(5)]
        assert plans["markov2-sparse"][0] == sorted({1 % size, 2 % size})
        assert plans["full"][0] == sorted({j % size for j in range(1, 5)})
        dm, dl, dh, _ = out["predict"]
        assert dm < 1e-12 and dl < 1e-12 and dh < 1e-12
    # every rank ends up with the same hyper-parameters and the same predictions, bit for bit
    for name in ("joint", "joint-tied"):
        assert all(results[r][name][3] == results[0][name][3] for r in range(size))
    assert all(results[r]["predict"][3] == results[0]["predict"][3] for r in range(size))
