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


def _pipeline_worker(rank, world, port, n_pairs, depth, steps, ret):
    """What one rank of `bench.py --gpus N` does with `depth` updates in flight (bench.py pipe_step): update i runs on slot
    i % depth, its local rows are gathered into THAT slot's table, and a slot's table is only rewritten `depth` updates later —
    so after every update the tables of the `depth` most recent updates must all be complete and intact."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = vdist.shard_range(n_pairs, rank, world)
    rows = torch.arange(b, e, dtype=torch.float64)[:, None] * 10.0 + torch.arange(6, dtype=torch.float64)[None, :]
    v_slots = [torch.zeros((e - b, 6), dtype=torch.float64) for _ in range(depth)]
    tables = [torch.zeros((n_pairs, 6), dtype=torch.float64) for _ in range(depth)]
    all_rows = torch.arange(n_pairs, dtype=torch.float64)[:, None] * 10.0 + torch.arange(6, dtype=torch.float64)[None, :]
    ok = True
    for i in range(steps):
        k = i % depth
        v_slots[k].copy_(rows + 1000.0 * i)                   # update i's twists of this rank's pairs
        vdist.gather_velocities(v_slots[k], n_pairs, out=tables[k])
        for back in range(min(depth, i + 1)):                 # the depth most recent updates, each in its own slot
            j = i - back
            ok = ok and bool(torch.equal(tables[j % depth], all_rows + 1000.0 * j))
    ret[rank] = (ok, tables[(steps - 1) % depth].clone())
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [8, 11])
def test_pipelined_gathers_on_eight_ranks_with_ragged_shards(n_pairs):
    """World size 8 (the node the driver scales to), even (8 pairs: BASELINE.json configs[3], one camera per GPU) and ragged
    (11 pairs: shards of 2, 2, 2, 1, 1, 1, 1, 1) shards, three slots round-robin as bench.py issues them."""
    world, depth, steps = 8, 3, 7
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_pipeline_worker, args=(world, port, n_pairs, depth, steps, ret), nprocs=world, join=True)
    want = torch.arange(n_pairs, dtype=torch.float64)[:, None] * 10.0 + torch.arange(6, dtype=torch.float64)[None, :] + 1000.0 * (steps - 1)
    for rank in range(world):
        ok, last = ret[rank]
        assert ok, f"rank {rank}: a slot's table was incomplete or overwritten early"
        assert torch.equal(last, want)
    sizes = [vdist.shard_range(n_pairs, r, world) for r in range(world)]
    assert sum(e - b for b, e in sizes) == n_pairs and max(e - b for b, e in sizes) - min(e - b for b, e in sizes) <= 1
