"""The N>1 layout (pair sharding + the v_c all-gather) on world_size-2 gloo, CPU only."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

import vitvs_amd  # noqa: F401
from vitvs_amd import dist as vdist


def test_shard_range_partitions():
    for n in (1, 2, 7, 8, 9, 64):
        for world in (1, 2, 3, 4, 8):
            spans = [vdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, n_pairs, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = vdist.shard_range(n_pairs, rank, world)
    v_local = torch.arange(b * 6, e * 6, dtype=torch.float64).reshape(e - b, 6) * 0.5
    full = vdist.gather_velocities(v_local, n_pairs)
    ret[rank] = full.clone()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [2, 8, 5])
def test_gather_velocities_two_ranks(n_pairs):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, n_pairs, ret), nprocs=2, join=True)
    want = torch.arange(n_pairs * 6, dtype=torch.float64).reshape(n_pairs, 6) * 0.5
    assert torch.equal(ret[0], want) and torch.equal(ret[1], want)


def _async_worker(rank, world, port, steps, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gather = vdist.VelocityGather(world, torch.device("cpu"))
    v = [torch.zeros((1, 6), dtype=torch.float64) for _ in range(2)]
    seen = []
    for i in range(steps):
        v[i & 1].fill_(100.0 * i + rank)          # this step's local result, in the slot the gather will read
        gather.post(v[i & 1], i)
        if i > 0:
            seen.append(gather.latest.clone())     # table of step i - 1 (post() completed it before issuing step i)
    seen.append(gather.finish().clone())
    ret[rank] = torch.stack(seen)
    dist.destroy_process_group()


def test_asynchronous_velocity_gather_two_ranks():
    """bench.py's N > 1 step: the gather of update i is waited for when update i + 1 posts its own; every table must be the
    complete one of its update on both ranks."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    steps = 7
    mp.spawn(_async_worker, args=(2, port, steps, ret), nprocs=2, join=True)
    for rank in (0, 1):
        got = ret[rank]
        assert got.shape == (steps, 2, 6)
        for i in range(steps):
            assert torch.all(got[i, 0] == 100.0 * i + 0) and torch.all(got[i, 1] == 100.0 * i + 1)
