"""Multi-GPU layout of the hot path: independent frame pairs sharded contiguously over ranks, one
process per GPU, weights replicated, and a single collective per update — the all-gather of the
6-double twists (48 bytes per pair; latency-bound, so a flat gather, never a ring of buckets).

The reference has no distributed code; this is the 8-camera-rig layout of
BASELINE.json configs[3].  ``torch.distributed`` backend "nccl" is RCCL on ROCm; the same code runs
on "gloo" for the CPU tests (tests/test_dist_gloo.py).
"""
from __future__ import annotations

from typing import Tuple

import torch


def shard_range(n_pairs: int, rank: int, world: int) -> Tuple[int, int]:
    """[begin, end) of the pairs rank ``rank`` owns: contiguous, sizes differ by at most one."""
    if not 0 <= rank < world:
        raise ValueError("rank outside the world")
    base, extra = divmod(n_pairs, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def gather_velocities(v_local: torch.Tensor, n_pairs: int, group=None, out: torch.Tensor = None) -> torch.Tensor:
    """All ranks receive v_c of every pair, [n_pairs, 6] float64, in global pair order.

    ``v_local`` is this rank's [n_local, 6] block (n_local from ``shard_range``).  Equal shards use
    one ``all_gather_into_tensor``; ragged shards are padded to the largest shard first."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(n_pairs, r, world) for r in range(world)]
    n_max = max(e - b for b, e in sizes)
    b, e = sizes[rank]
    if v_local.shape != (e - b, 6):
        raise ValueError(f"rank {rank} should hold {(e - b, 6)}, got {tuple(v_local.shape)}")
    if v_local.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal path (several ranks sharing one GPU, gloo): stage the 48-byte rows through the host
        full_cpu = gather_velocities(v_local.detach().cpu(), n_pairs, group)
        if out is not None:
            out.copy_(full_cpu)
            return out
        return full_cpu.to(v_local.device)
    if all(e2 - b2 == n_max for b2, e2 in sizes):
        full = out if out is not None else torch.empty((n_pairs, 6), dtype=v_local.dtype, device=v_local.device)
        dist.all_gather_into_tensor(full, v_local.contiguous(), group=group)
        return full
    padded = torch.zeros((n_max, 6), dtype=v_local.dtype, device=v_local.device)
    padded[: e - b] = v_local
    buf = torch.empty((world * n_max, 6), dtype=v_local.dtype, device=v_local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    full = out if out is not None else torch.empty((n_pairs, 6), dtype=v_local.dtype, device=v_local.device)
    for r, (b2, e2) in enumerate(sizes):
        full[b2:e2] = buf[r * n_max: r * n_max + (e2 - b2)]
    return full


class VelocityGather:
    """The per-update ``v_c`` all-gather issued asynchronously (opt-in: ``bench.py`` with VITVS_ASYNC_GATHER=1).

    Measured on one MI355X in a world of one rank (round 1): slower than the synchronous gather (0.577 vs 0.464 ms per
    update) — the extra queue's events cost more than the wait they remove; kept for multi-GPU experiments, where the
    collective's latency is longer.  With several updates in flight (vit-vs_amd/pipeline.py) each update's gather simply
    follows it on its own stream and overlaps the other updates.

    ``post(v_local)`` issues the collective asynchronously: with RCCL it runs on the communicator's own stream behind an
    event of the caller's stream, so the next update's launches do not wait for it; the previous update's collective is
    waited for first (it finished long before), and results alternate between two output buffers, so a buffer is never
    rewritten while a collective may still touch it.  The caller alternates its ``v_local`` buffers the same way
    (``slot`` = update index & 1).  ``finish()`` waits for the outstanding collective; ``latest`` is the last complete
    [n_pairs, 6] table.  Equal shards only (the benchmark's layout); ragged shards use ``gather_velocities``.
    """

    def __init__(self, n_pairs: int, device, dtype=torch.float64, group=None):
        import torch.distributed as dist
        self.group = group
        world = dist.get_world_size(group)
        if n_pairs % world != 0:
            raise ValueError("VelocityGather needs equal shards")
        self.n_pairs = n_pairs
        self.out = [torch.zeros((n_pairs, 6), dtype=dtype, device=device) for _ in range(2)]
        self.pending = None
        self.pending_slot = -1
        self.latest = self.out[0]

    def post(self, v_local: torch.Tensor, slot: int):
        import torch.distributed as dist
        self.finish()
        self.pending = dist.all_gather_into_tensor(self.out[slot & 1], v_local, group=self.group, async_op=True)
        self.pending_slot = slot & 1

    def finish(self):
        if self.pending is not None:
            self.pending.wait()
            self.latest = self.out[self.pending_slot]
            self.pending = None
        return self.latest
