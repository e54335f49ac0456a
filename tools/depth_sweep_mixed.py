"""Updates in flight beyond the four hardware queues of one priority class: slots alternate between the high-priority and the
default class (HIP hands out GPU_MAX_HW_QUEUES = 4 queues per class).  ViT-B/16 224², one pair per update, bf16.
usage: python tools/depth_sweep_mixed.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vitvs_amd  # noqa: F401
from vitvs_amd import _lib, config, synth, weights
from vitvs_amd.pipeline import UpdatePipeline


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    dev = torch.device("cuda", 0)
    cfg = config.baseline_config("vitb16_224")
    params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=False)
    sd = weights.synthetic_state_dict(cfg, 0)
    des, cur = synth.frame_pair(cfg.img_size, synth.ACCEPTED_FRAME_SEEDS["vitb16_224"])
    des_d, cur_d = torch.from_numpy(des[None]).to(dev), torch.from_numpy(cur[None]).to(dev)
    Z = torch.from_numpy(synth.depth_pattern()[None]).to(dev)
    K = torch.tensor([params.intrinsics()], dtype=torch.float64, device=dev)
    order = torch.randperm(cfg.tokens, generator=torch.Generator().manual_seed(121)).to(torch.int32).to(dev)[None]
    for depth, mixed in ((3, False), (4, False), (4, True), (5, True), (6, True), (8, True), (3, False)):
        pipe = UpdatePipeline(cfg, params, sd, precision="bf16", depth=depth)
        if mixed:
            pipe.streams = [torch.cuda.Stream(device=dev, priority=(-1 if k % 2 == 0 else 0)) for k in range(depth)]
        for _ in range(4 * depth):
            pipe.submit(cur_d, des_d, Z, K, _lib.SELECT_ORDER, order)
        pipe.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.submit(cur_d, des_d, Z, K, _lib.SELECT_ORDER, order)
        pipe.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"depth {depth} {'high / default classes alternating' if mixed else 'high-priority class only':36s} "
              f"{steps / dt:8.1f} updates/s", flush=True)
        pipe.close()


if __name__ == "__main__":
    main()
