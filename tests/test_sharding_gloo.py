"""CPU tests of the N > 1 path: agent-range sharding over torch.distributed (gloo, world_size 2 and 3).

bench.py launches one process per GPU with the same helpers over RCCL; here the processes run on CPU
tensors so the logic (ranges, scatter/gather, neighbour-state all-gather, max-over-ranks timing,
shared obstacle table broadcast) is exercised without a GPU.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from safe_control_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_agents, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = sharding.agent_range(n_agents, world, rank)
        full = torch.arange(n_agents * 4, dtype=torch.float64).reshape(n_agents, 4) if rank == 0 else None
        # scatter: every rank gets its contiguous rows
        local = sharding.scatter_agents(full if rank == 0 else torch.empty(0), n_agents, src=0)
        want = torch.arange(n_agents * 4, dtype=torch.float64).reshape(n_agents, 4)[lo:hi]
        assert torch.equal(local, want), (rank, local, want)
        # "solve" locally (no collective on the data path), then gather on rank 0
        res = local * 2.0 + rank * 0.0
        back = sharding.gather_agents(res, n_agents, dst=0)
        if rank == 0:
            assert torch.equal(back, torch.arange(n_agents * 4, dtype=torch.float64).reshape(n_agents, 4) * 2.0)
        else:
            assert back is None
        # neighbour-state exchange: everyone sees every agent's state, in agent order
        allx = sharding.all_gather_states(local, n_agents)
        assert torch.equal(allx, torch.arange(n_agents * 4, dtype=torch.float64).reshape(n_agents, 4))
        # the persistent-buffer exchange used by bench.py's collective leg: same result, equal and ragged shards, repeated steps
        ex = sharding.NeighborExchange(n_agents, 4, 0.3, nx=4, dtype=torch.float64, device="cpu")
        assert ex.equal == (n_agents % world == 0)
        for rep in range(2):
            got = ex.gather(local + rep)
            assert torch.equal(got, torch.arange(n_agents * 4, dtype=torch.float64).reshape(n_agents, 4) + rep)
            assert got.data_ptr() == ex.X_all.data_ptr()                       # written in place, no new allocation
        # shared obstacle table
        table = torch.full((5, 7), float(rank))
        sharding.broadcast_obstacle_table(table, src=0)
        assert torch.all(table == 0.0)
        # timing contract: max over ranks
        t = sharding.max_over_ranks(1.0 + rank)
        assert t == float(world)
        q.put((rank, lo, hi))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_agents", [(2, 4096), (3, 10)])
def test_sharded_pipeline_gloo(world, n_agents):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_agents, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    # ranges tile [0, n) without gaps or overlap
    assert got[0][1] == 0 and got[-1][2] == n_agents
    for a, b in zip(got, got[1:]):
        assert a[2] == b[1]


def test_agent_range_properties():
    for n in (0, 1, 5, 64, 4097, 65536):
        for w in (1, 2, 3, 8):
            sizes = sharding.shard_sizes(n, w)
            assert sum(sizes) == n and max(sizes) - min(sizes) <= 1
            ends = [sharding.agent_range(n, w, r) for r in range(w)]
            assert ends[0][0] == 0 and ends[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ends, ends[1:]))


def test_single_process_is_a_no_op():
    x = torch.arange(12.0).reshape(3, 4)
    assert sharding.scatter_agents(x, 3) is x and sharding.gather_agents(x, 3) is x
    assert sharding.all_gather_states(x, 3) is x and sharding.max_over_ranks(2.5) == 2.5
