"""The asynchronous v_c gather on the RCCL backend (torch.distributed "nccl"), as far as one GPU allows: a world of one
rank exercises the same calls bench.py makes at N > 1 (all_gather_into_tensor with async_op, wait on the next post,
alternating buffers) on a side stream with device tensors."""
import os
import socket

import pytest
import torch

import vitvs_amd  # noqa: F401
from vitvs_amd import dist as vdist

pytestmark = pytest.mark.gpu


def test_velocity_gather_on_rccl_single_rank():
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            gather = vdist.VelocityGather(1, dev)
            v = [torch.zeros((1, 6), dtype=torch.float64, device=dev) for _ in range(2)]
            tables = []
            for i in range(6):
                v[i & 1].fill_(float(i) + 0.25)          # stands for the servo kernel writing this update's v_c
                gather.post(v[i & 1], i)
                if i > 0:
                    tables.append(gather.latest.clone())
            tables.append(gather.finish().clone())
            torch.cuda.synchronize(dev)
            dist.barrier()
        for i, t in enumerate(tables):
            assert torch.all(t.cpu() == float(i) + 0.25)
    finally:
        dist.destroy_process_group()
