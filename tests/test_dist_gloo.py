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
