"""N > 1 path on CPU: two gloo ranks run the Monte-Carlo data path of `contactimplicitmpc/jl_amd/monte_carlo.py` - rank 0
builds the problem, the packed linearization tables / objective are BROADCAST, every rank solves its shard of the
rollouts with a REAL solve (the CPU checker's newton_solve!, oracle/cimpc_ref.c, standing in for the device), one
ALL-GATHER returns [u1 | newton_iters | r_norm | sweeps] in global order.  The gathered result must be bit-identical to
the unsharded run: rollouts are independent, so sharding must not change any result.
The same functions run over RCCL in bench.py --gpus N."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from contactimplicitmpc.jl_amd import synthetic as synth
from contactimplicitmpc.jl_amd.monte_carlo import problem_shapes, run_monte_carlo
from contactimplicitmpc.jl_amd.sharding import rollout_shard
from contactimplicitmpc.jl_amd.trajectory import Dims, Objective, QUADRUPED

H, H_REF = 8, 12


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _dims():
    return Dims(**QUADRUPED)


def _build():
    d = _dims()
    prob = synth.make_problem(d, H_REF, seed=1)
    obj = synth.make_objective(d, H)
    return prob, obj.q, obj.u


def _rollouts(prob_full, first, count):
    d = _dims()
    out = []
    for g in range(first, first + count):          # GLOBAL index: the batch does not depend on the sharding
        phase = int(np.random.default_rng(7919 + g).integers(0, H_REF))
        out.append(synth.make_rollout(d, prob_full, H, phase=phase, seed=100003 + g, perturb=0.02))
    return out


def _solve(prob, obj_q, obj_u, rollouts):
    """Per-shard solve with the CPU checker (the device solver has the same signature in bench.py)."""
    from oracle import ip as oip, newton as onewton
    from oracle.cref import CRef
    d = _dims()
    cr = CRef(d, H_REF, H, prob, Objective(q=obj_q, u=obj_u), oip.IPOptions(kappa_tol=prob["kappa"]),
              onewton.NewtonOptions(r_tol=1e-5, max_iter=4), prob["kappa"])
    res = [cr.newton_solve(w, ref, q0, q1, solver=1) for (w, ref, q0, q1) in rollouts]
    return (np.stack([r["u"][0] for r in res]), np.array([r["iters"] for r in res]),
            np.array([r["r_norm"] for r in res]), np.array([r["sweeps"] for r in res]))


def _shapes():
    d = _dims()
    return problem_shapes(H_REF, H, d.nq, d.nu, d.nw, d.nc, d.nb)


def _worker(rank, world, port, n_rollouts, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # only rank 0 builds the problem; linearization tables, reference gait and objective reach rank 1 by broadcast
    res = run_monte_carlo(n_rollouts, _shapes(), _build, _rollouts, _solve, device="cpu")
    np.savez(out_path + ".%d.npz" % rank, **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_rollouts", [6, 5])
def test_two_rank_shard_equals_unsharded(tmp_path, n_rollouts):
    port = _free_port()
    out = str(tmp_path / "gathered")
    mp.spawn(_worker, args=(2, port, n_rollouts, out), nprocs=2, join=True)
    prob, oq, ou = _build()
    ref = _solve(prob, oq, ou, _rollouts(prob, 0, n_rollouts))
    for rank in (0, 1):         # every rank holds the full gathered result
        got = np.load(out + ".%d.npz" % rank)
        assert got["u1"].shape == ref[0].shape
        assert np.array_equal(got["u1"], ref[0])
        assert np.array_equal(got["newton_iters"], ref[1])
        assert np.array_equal(got["r_norm"], ref[2])
        assert np.array_equal(got["sweeps"], ref[3])
    assert ref[1].max() >= 1          # the shard solves did real Newton iterations


def test_shard_partition_properties():
    for n in (1, 7, 512, 513):
        for world in (1, 2, 4, 8):
            spans = [rollout_shard(n, r, world) for r in range(world)]
            assert sum(c for _, c in spans) == n
            pos = 0
            for s, c in spans:
                assert s == pos
                pos += c
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
