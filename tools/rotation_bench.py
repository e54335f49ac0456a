"""Throughput of the rotation search's call shape (find_and_set_best_pose, vitvs_v2.py:1151-1189): four candidate views against
ONE goal image in a single call (des_shared), 48 feature pairs per view, a fresh visiting order per call, device-resident inputs.
    python tools/rotation_bench.py [config key] [precision]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vitvs_amd  # noqa: E402,F401
from vitvs_amd import _lib, config, synth, weights  # noqa: E402
from vitvs_amd.engine import Engine  # noqa: E402


def main():
    key = sys.argv[1] if len(sys.argv) > 1 else "vitb16_224"
    prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    cfg = config.baseline_config(key)
    params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=False)
    dev = torch.device("cuda")
    eng = Engine(cfg, params, precision=prec, max_pairs=4, max_rows=48).load_state_dict(weights.synthetic_state_dict(cfg, 0))
    des, cur0 = synth.frame_pair(cfg.img_size, 20250705)
    views = np.stack([np.rot90(cur0, k).copy() for k in range(4)])                # the four orientations of the search
    I_des = torch.from_numpy(des[None]).to(dev)
    I_cur = torch.from_numpy(views).to(dev)
    Z = torch.from_numpy(np.stack([synth.depth_pattern()] * 4)).to(dev)
    K = torch.tensor([params.intrinsics()] * 4, dtype=torch.float64, device=dev)
    gen = torch.Generator().manual_seed(3)
    orders = torch.stack([torch.stack([torch.randperm(cfg.tokens, generator=gen) for _ in range(4)]) for _ in range(64)]).to(torch.int32).to(dev)
    v = torch.zeros((4, 6), dtype=torch.float64, device=dev)
    st = torch.zeros(4, dtype=torch.int32, device=dev)

    def call(i, goal):
        eng.compute_velocity_dev(I_cur, goal, Z, K, _lib.SELECT_ORDER, orders[i % 64], None, True, v, st, num_pairs=48)

    for label, goal in (("goal forwarded in every call (the reference's behaviour)", I_des), ("goal cached (Engine.set_goal)", None)):
        if goal is None:
            eng.set_goal(I_des)
        for i in range(10):
            call(i, goal)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 200
        for i in range(n):
            call(i, goal)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{key} {prec}: rotation search call (4 views, 1 goal, 48 pairs each), {label}: {dt * 1e3:.3f} ms = {1 / dt:.0f} searches/s")


if __name__ == "__main__":
    main()
