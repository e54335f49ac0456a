"""Soak of the pipelined path (tools only): many thousand updates through a pipeline (depth 4 = bench.py's default, or argv[4]), every
result compared with the first pass over the same (frame pair, visiting order) — run-to-run bit reproducibility with every queue busy —
and the device memory in use before and after.

  python tools/soak_pipeline.py [updates] [config key] [binned|-] [depth]
"""
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
    depth = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    npairs = int(os.environ.get("SOAK_PAIRS", 8 if depth % 2 else 7))   # coprime with the depth: every pair meets every slot
    dev = torch.device("cuda", 0)
    cfg = config.baseline_config(key)
    params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=binned)
    sd = weights.synthetic_state_dict(cfg, 0)
    seeds = synth.RIG8_FRAME_SEEDS if key == "vitb16_224" else [synth.ACCEPTED_FRAME_SEEDS[key] + i for i in range(8)]
    pairs = [synth.frame_pair(cfg.img_size, s) for s in seeds[:npairs]]
    des = [torch.from_numpy(p[0][None]).to(dev) for p in pairs]
    cur = [torch.from_numpy(p[1][None]).to(dev) for p in pairs]
    Z = torch.from_numpy(synth.depth_pattern()[None]).to(dev)
    K = torch.tensor([params.intrinsics()], dtype=torch.float64, device=dev)
    gen = torch.Generator().manual_seed(121)
    P = 5 * npairs                                      # period of the (pair, order) schedule: coprime with the depth
    orders = torch.stack([torch.randperm(cfg.tokens, generator=gen) for _ in range(P)]).to(torch.int32).to(dev)[:, None]
    pipe = UpdatePipeline(cfg, params, sd, precision="bf16", depth=depth)
    free0, reserved0 = torch.cuda.mem_get_info(dev)[0], torch.cuda.memory_reserved(dev)
    free_warm = reserved_warm = None
    first, bad, tickets = {}, 0, []
    t0 = time.perf_counter()
    for i in range(n):
        if i == 3 * P:
            # every (slot, frame pair) has captured and instantiated its graph (depth x pairs of them: a hipGraphExec owns its kernel-argument
            # and node memory on the device) and torch's caching allocator has created its pools (the results' .clone(): one 2 MiB
            # small-block segment; any allocation over 1 MiB reserves a 20 MiB segment): what is in use from here on must not grow
            torch.cuda.synchronize(dev)
            free_warm, reserved_warm = torch.cuda.mem_get_info(dev)[0], torch.cuda.memory_reserved(dev)
        tickets.append((i, pipe.submit(cur[i % npairs], des[i % npairs], Z, K, _lib.SELECT_ORDER, orders[i % P])))
        if len(tickets) == depth:
            j, t = tickets.pop(0)
            v = pipe.result(t)[0].cpu().numpy().tobytes()
            if first.setdefault(j % P, v) != v:
                bad += 1
    for j, t in tickets:
        v = pipe.result(t)[0].cpu().numpy().tobytes()
        bad += first.setdefault(j % P, v) != v
    dt = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    free1, reserved1 = torch.cuda.mem_get_info(dev)[0], torch.cuda.memory_reserved(dev)
    mib = 2.0 ** 20
    steady = "n/a (run shorter than the warm-up)" if free_warm is None else \
        f"{(free_warm - free1) / mib:+.1f} MiB over the {n - 3 * P} updates after it (torch's allocator {(reserved1 - reserved_warm) / mib:+.1f} MiB)"
    warm = "" if free_warm is None else \
        f"{(free0 - free_warm) / mib:+.1f} MiB during the first {3 * P} updates (graph capture + instantiation of {depth * npairs} (slot, pair) graphs, " \
        f"of which torch's caching allocator reserved {(reserved_warm - reserved0) / mib:+.1f} MiB), "
    print(f"{key}{' binned' if binned else ''}: {n} updates through {depth} slots in {dt:.2f} s ({n / dt:.0f} updates/s with a host read per update), "
          f"{len(first)} distinct (pair, order) cases, results differing from their first pass: {bad}, "
          f"device memory in use: {warm}{steady}")
    leak = free_warm is not None and (free_warm - free1) > 2 * mib
    sys.exit(1 if bad or leak else 0)


if __name__ == "__main__":
    main()
