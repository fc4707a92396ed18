"""Batch-sharded Monte-Carlo rollouts over the GPUs of one node (BASELINE configs[3]).

The reference runs its Monte-Carlo study as a serial loop over initial conditions
(/root/reference/examples/quadruped/monte_carlo.jl:76-92: one `simulate!` per sample, every sample with the same
linearization / objective / gait).  Here the samples are independent ROLLOUTS of one batched solve:

  rank 0 builds the shared problem (per-knot linearization z0, th0, r0, rz0, rth0 + objective blocks), packs it into ONE
  contiguous tensor and broadcasts it (RCCL over xGMI: `dist.broadcast`, 1.3 MB for the quadruped's 60 knots);
  every rank takes its contiguous shard of the global rollout indices (`sharding.rollout_shard`), generates / receives
  that shard's initial conditions and runs the solve on its own GPU - NO collective inside a solve;
  per reporting interval ONE all-gather returns `[u1 | newton_iters | r_norm | sweeps]` per rollout in global order.

The same code runs on CPU ranks (gloo) with any per-shard `solve_fn` - the world_size-2 test plugs the CPU checker's
newton_solve! in, the bench and the scripts the device solver.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .sharding import allgather_rollouts, broadcast_tables, rollout_shard

_FIELDS = ("z0", "th0", "r0", "rz0", "rth0", "q_ref", "u_ref", "w_ref", "gamma_ref", "b_ref")


def _world():
    on = dist.is_available() and dist.is_initialized()
    return (dist.get_rank(), dist.get_world_size()) if on else (0, 1)


def pack_problem(prob, obj_q, obj_u):
    """The shared read-only data of a Monte-Carlo batch - per-knot linearization, the reference gait the rollouts
    start from, the objective blocks, kappa - as one float64 vector."""
    parts = [np.ascontiguousarray(prob[k], dtype=np.float64).ravel() for k in _FIELDS]
    parts += [np.ascontiguousarray(obj_q, dtype=np.float64).ravel(), np.ascontiguousarray(obj_u, dtype=np.float64).ravel(),
              np.array([prob["kappa"]], dtype=np.float64)]
    return np.concatenate(parts)


def problem_shapes(H_ref, H, nq, nu, nw, nc, nb):
    nz, nth = nq + 4 * nc + 2 * nb, 2 * nq + nu + nw + 2
    return [(H_ref, nz), (H_ref, nth), (H_ref, nz), (H_ref, nz, nz), (H_ref, nz, nth),
            (H_ref + 2, nq), (H_ref, nu), (H_ref, nw), (H_ref, nc), (H_ref, nb), (H, nq, nq), (H, nu, nu), (1,)]


def unpack_problem(vec, shapes):
    out, off = [], 0
    for sh in shapes:
        n = int(np.prod(sh))
        out.append(np.asarray(vec[off:off + n]).reshape(sh).copy())
        off += n
    assert off == len(vec)
    prob = dict(zip(_FIELDS, out[:len(_FIELDS)]))
    prob["kappa"] = float(out[-1][0])
    return prob, out[-3], out[-2]


def broadcast_problem(prob, obj_q, obj_u, shapes, device):
    """Rank 0 passes the problem, the other ranks pass None; everybody returns (prob, obj_q, obj_u) - bit-identical."""
    rank, world = _world()
    n = sum(int(np.prod(s)) for s in shapes)
    if rank == 0:
        t = torch.from_numpy(pack_problem(prob, obj_q, obj_u)).to(device)
        assert t.numel() == n
    else:
        t = torch.empty(n, dtype=torch.float64, device=device)
    broadcast_tables(t, src=0)
    if world > 1 and t.is_cuda:
        torch.cuda.synchronize()
    return unpack_problem(t.cpu().numpy(), shapes)


def gather_results(u1, newton_iters, r_norm, sweeps, n_rollouts, device):
    """One all-gather per reporting interval: rows of [u1 (nu) | newton_iters | r_norm | sweeps] in global rollout order."""
    loc = np.concatenate([np.asarray(u1, dtype=np.float64), np.asarray(newton_iters, dtype=np.float64)[:, None],
                          np.asarray(r_norm, dtype=np.float64)[:, None], np.asarray(sweeps, dtype=np.float64)[:, None]], axis=1)
    g = allgather_rollouts(torch.from_numpy(loc).to(device), n_rollouts).cpu().numpy()
    nu = g.shape[1] - 3
    return dict(u1=g[:, :nu], newton_iters=g[:, nu].astype(np.int64), r_norm=g[:, nu + 1], sweeps=g[:, nu + 2].astype(np.int64))


def run_monte_carlo(n_rollouts, shapes, build_problem, make_rollouts, solve_fn, device="cpu"):
    """The whole N-rank data path of one reporting interval.

    build_problem()              -> (prob, obj_q, obj_u); called on rank 0 only
    make_rollouts(prob, first, count) -> the rank's shard of initial conditions (global indices first .. first+count-1)
    solve_fn(prob, obj_q, obj_u, rollouts) -> (u1 (count, nu), newton_iters, r_norm, sweeps) for the shard
    Returns the gathered dict (every rank holds the full result)."""
    rank, world = _world()
    prob, obj_q, obj_u = build_problem() if rank == 0 else (None, None, None)
    prob, obj_q, obj_u = broadcast_problem(prob, obj_q, obj_u, shapes, device)
    first, count = rollout_shard(n_rollouts, rank, world)
    ro = make_rollouts(prob, first, count)
    u1, it, rn, sw = solve_fn(prob, obj_q, obj_u, ro)
    return gather_results(u1, it, rn, sw, n_rollouts, device)
