"""The entry point the round-end driver uses for N > 1: `python bench.py --gpus N` with no launcher around it must start the
N ranks itself (one process per GPU, 127.0.0.1, a free port), and `python -m torch.distributed.run ... bench.py --gpus N` must
keep working.  `--launch-check` stops after the process group has formed and answered one collective (gloo: no GPU needed),
so the launcher is exercised here on the CPU; the timed path behind it needs the MI355X."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _last_json(stdout):
    lines = [line for line in stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, stdout  # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("gpus", [1, 2, 4])
def test_plain_invocation_starts_its_own_ranks(gpus):
    out = subprocess.run([sys.executable, BENCH, "--gpus", str(gpus), "--launch-check"], env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = _last_json(out.stdout)
    assert line["n_gpus"] == gpus and line["sum_of_ranks_plus_one"] == gpus * (gpus + 1) / 2


def test_invocation_under_torchrun_as_the_driver_does_it():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--launch-check"]
    out = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert _last_json(out.stdout)["n_gpus"] == 2


def test_world_size_that_contradicts_the_flag_is_refused():
    env = dict(_env(), WORLD_SIZE="3", RANK="0")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=3" in (out.stderr + out.stdout)


def test_start_up_watchdog_ends_a_rank_whose_peers_never_arrive():
    """Rank 0 of a 2-rank group whose second rank does not exist: the process must leave by itself (exit code 3 and an error
    line instead of a hang)."""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check", "--init-timeout", "5"], env=env, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 3, (out.returncode, out.stderr[-1000:])
    assert "did not form" in _last_json(out.stdout)["error"]


@pytest.mark.gpu
def test_two_ranks_on_the_hip_engine_agree_with_one():
    """The multi-rank control flow on the REAL engine: `python bench.py --gpus 2` starts two ranks itself; with GPAR_BENCH_ONE_GPU=1
    (a development mode: both ranks use GPU 0 and talk over gloo - timings mean nothing) they shard the layers of a small model,
    all-reduce the layer log-likelihoods and run the sharded fit + predict leg; the log marginal likelihood must equal the
    single-process value to rounding."""
    def run(gpus, extra_env):
        cmd = [sys.executable, BENCH, "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--rows", "1536", "--p", "4", "--no-cpu"]
        out = subprocess.run(cmd, env=dict(_env(), **extra_env), capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        return _last_json(out.stdout)

    one = run(1, {})
    two = run(2, {"GPAR_BENCH_ONE_GPU": "1"})
    assert two["n_gpus"] == 2 and two["config"]["layers_per_rank"] == [2, 2] and len(two["per_rank_busy_ms"]) == 2
    assert abs(two["config"]["logpdf"] - one["config"]["logpdf"]) <= 1e-10 * abs(one["config"]["logpdf"])
    for line in (one, two):   # the sharded fit (layer pi trained on rank pi mod 2, latents broadcast) and sample-parallel predict ran
        assert "error" not in line["fit_predict"], line["fit_predict"]
        assert line["fit_predict"]["fit_evaluations"] > 20 and np.isfinite(line["fit_predict"]["predict_mean_abs"])


_RCCL_PROBE = r"""
import datetime, os, sys
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=60), device_id=torch.device("cuda:0"))
dev = torch.device("cuda:0")
word = torch.full((1,), 3.25, dtype=torch.float64, device=dev)
dist.all_reduce(word, op=dist.ReduceOp.SUM)                      # sharded_logpdf: sum of the layer log-likelihoods
peak = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(peak, op=dist.ReduceOp.MAX)                      # bench.py: max over ranks of the elapsed time
col = torch.arange(5.0, dtype=torch.float64, device=dev).reshape(5, 1)
dist.broadcast(col, src=0)                                       # dependent regimes: the forwarded column
send = torch.arange(6.0, dtype=torch.float64, device=dev)
recv = [torch.empty_like(send)]
dist.all_gather(recv, send)                                      # packed factors / samples
dist.barrier()
torch.cuda.synchronize()
assert float(word) == 3.25 and float(peak) == 1.5 and torch.equal(recv[0], send) and float(col.sum()) == 10.0
dist.destroy_process_group()
print("rccl ok")
"""


@pytest.mark.gpu
def test_rccl_forms_a_communicator_and_carries_fp64_on_this_box(tmp_path):
    """What `bench.py --gpus N` and gpar_amd/parallel.py ask of RCCL, on the one GPU of the test box: a communicator bound to the
    device at init (`device_id=`), and the collectives they issue on fp64 device tensors.  A world of one rank moves no data
    between GPUs - it shows that the backend loads and accepts these calls (the N > 1 control flow runs over gloo above and in
    tests/test_distributed.py); the 8-GPU run itself is the driver's."""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    script = tmp_path / "rccl_probe.py"
    script.write_text(_RCCL_PROBE)
    out = subprocess.run([sys.executable, str(script), str(port)], env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "rccl ok" in out.stdout, out.stderr[-3000:]
