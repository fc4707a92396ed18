"""N > 1 path on CPU: two gloo ranks shard a rollout batch, 'solve' their shard independently
and all-gather; the gathered result must be bit-identical to the unsharded run (rollouts are
independent, so sharding must not change any result - SURVEY.md section 4/8e)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from contactimplicitmpc.jl_amd.sharding import allgather_rollouts, broadcast_tables, rollout_shard


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _solve_stub(q0, table):
    """Stand-in for the per-rollout solve: any deterministic per-row function."""
    return torch.tanh(q0 @ table) + q0.sum(dim=1, keepdim=True)


def _worker(rank, world, port, n_rollouts, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(7)
    q0_all = torch.randn(n_rollouts, 11, generator=g, dtype=torch.float64)
    table = torch.randn(11, 8, generator=g, dtype=torch.float64) if rank == 0 else torch.zeros(11, 8, dtype=torch.float64)
    broadcast_tables(table, src=0)
    start, count = rollout_shard(n_rollouts, rank, world)
    local = _solve_stub(q0_all[start:start + count], table)
    full = allgather_rollouts(local, n_rollouts)
    if rank == 0:
        np.save(out_path, full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_rollouts", [8, 7])
def test_two_rank_shard_equals_unsharded(tmp_path, n_rollouts):
    port = _free_port()
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, port, n_rollouts, out), nprocs=2, join=True)
    g = torch.Generator().manual_seed(7)
    q0_all = torch.randn(n_rollouts, 11, generator=g, dtype=torch.float64)
    table = torch.randn(11, 8, generator=g, dtype=torch.float64)
    ref = _solve_stub(q0_all, table).numpy()
    got = np.load(out)
    assert got.shape == ref.shape and np.array_equal(got, ref)


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
