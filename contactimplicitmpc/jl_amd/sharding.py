"""Multi-GPU layout of the path: Monte-Carlo rollouts are independent
(/root/reference/examples/quadruped/monte_carlo.jl:83-91 runs them in a serial `for`), so the
batch is sharded over ranks (one process per GPU) and NO collective runs inside a solve.
Collectives (RCCL on GPUs, gloo in the CPU tests) are used only to
  * broadcast the shared linearization tables / objective once at setup, and
  * all-gather the per-rollout results (u[1], Newton iterations, residual norms) per
    reporting interval.
A horizon is never split across GPUs: the KKT solve couples all of its steps.
"""
import torch
import torch.distributed as dist


def rollout_shard(n_rollouts: int, rank: int, world: int):
    """Contiguous near-equal split of global rollout indices: returns (start, count)."""
    base, rem = divmod(n_rollouts, world)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def broadcast_tables(t: torch.Tensor, src: int = 0):
    """One-time broadcast of the packed linearization tables / objective blocks."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return t


def allgather_rollouts(local: torch.Tensor, n_rollouts: int):
    """All-gather a per-rollout tensor (leading dim = local rollout count) into the global
    order defined by rollout_shard.  Shards may differ by one row: pad to the max count."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [rollout_shard(n_rollouts, r, world)[1] for r in range(world)]
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
