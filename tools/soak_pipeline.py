"""Soak of the pipelined path (tools only): many thousand updates through a depth-3 pipeline, every result compared with the
first pass over the same (frame pair, visiting order) — run-to-run bit reproducibility under three queues — and the device
memory in use before and after."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vitvs_amd  # noqa: F401
from vitvs_amd import _lib, config, synth, weights
from vitvs_amd.pipeline import UpdatePipeline


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
    key = sys.argv[2] if len(sys.argv) > 2 else "vitb16_224"      # e.g. vitb8_448: the key-split long-sequence attention
    binned = len(sys.argv) > 3 and sys.argv[3] == "binned"        # the stencil form of the binned Gram (raw Gram workspace per slot)
    dev = torch.device("cuda", 0)
    cfg = config.baseline_config(key)
    params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=binned)
    sd = weights.synthetic_state_dict(cfg, 0)
    seeds = synth.RIG8_FRAME_SEEDS if key == "vitb16_224" else [synth.ACCEPTED_FRAME_SEEDS[key] + i for i in range(8)]
    pairs = [synth.frame_pair(cfg.img_size, s) for s in seeds]
    des = [torch.from_numpy(p[0][None]).to(dev) for p in pairs]
    cur = [torch.from_numpy(p[1][None]).to(dev) for p in pairs]
    Z = torch.from_numpy(synth.depth_pattern()[None]).to(dev)
    K = torch.tensor([params.intrinsics()], dtype=torch.float64, device=dev)
    gen = torch.Generator().manual_seed(121)
    P = 40                                              # period of the (pair, order) schedule: coprime with the depth
    orders = torch.stack([torch.randperm(cfg.tokens, generator=gen) for _ in range(P)]).to(torch.int32).to(dev)[:, None]
    pipe = UpdatePipeline(cfg, params, sd, precision="bf16", depth=3)
    free0 = torch.cuda.mem_get_info(dev)[0]
    first, bad, tickets = {}, 0, []
    t0 = time.perf_counter()
    for i in range(n):
        tickets.append((i, pipe.submit(cur[i % 8], des[i % 8], Z, K, _lib.SELECT_ORDER, orders[i % P])))
        if len(tickets) == 3:
            j, t = tickets.pop(0)
            v = pipe.result(t)[0].cpu().numpy().tobytes()
            if first.setdefault(j % P, v) != v:
                bad += 1
    for j, t in tickets:
        v = pipe.result(t)[0].cpu().numpy().tobytes()
        bad += first.setdefault(j % P, v) != v
    dt = time.perf_counter() - t0
    free1 = torch.cuda.mem_get_info(dev)[0]
    print(f"{key}{' binned' if binned else ''}: {n} updates through 3 slots in {dt:.2f} s ({n / dt:.0f} updates/s with a host read per update), "
          f"{len(first)} distinct (pair, order) cases, results differing from their first pass: {bad}, "
          f"device memory in use changed by {(free0 - free1) / 2**20:.1f} MiB")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
