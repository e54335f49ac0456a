"""Updates/s of one precision at one frame pair, one update in flight and three (the two regimes bench.py reports), plus the
per-class kernel times of one instrumented update — the short form used while iterating on kernels (bench.py is the record).

  python tools/quick_rate.py [--precision f16x2] [--config vitb16_224] [--steps 200] [--depth 3]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vitvs_amd  # noqa: E402,F401
from vitvs_amd import _lib, config, synth, weights  # noqa: E402
from vitvs_amd.engine import Engine  # noqa: E402
from vitvs_amd.pipeline import UpdatePipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f16x2")
    ap.add_argument("--config", default="vitb16_224")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=1)
    ap.add_argument("--no-hint", action="store_true", help="pipeline slots keep the one-stream tile plan (no in_flight hint)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = config.baseline_config(args.config)
    params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=False)
    sd = weights.synthetic_state_dict(cfg, 0)
    B = args.pairs
    des, cur = synth.frame_pair(cfg.img_size, synth.RIG8_FRAME_SEEDS[0] if args.config == "vitb16_224" else synth.ACCEPTED_FRAME_SEEDS[args.config])
    I_des = torch.from_numpy(np.stack([des] * B)).to(dev)
    I_cur = torch.from_numpy(np.stack([cur] * B)).to(dev)
    Z = torch.from_numpy(np.stack([synth.depth_pattern()] * B)).to(dev)
    K = torch.tensor([params.intrinsics()] * B, dtype=torch.float64, device=dev)
    gen = torch.Generator().manual_seed(121)
    n = args.steps
    orders = torch.stack([torch.stack([torch.randperm(cfg.tokens, generator=gen) for _ in range(B)]) for _ in range(32)]).to(torch.int32).to(dev)
    eng = Engine(cfg, params, precision=args.precision, max_pairs=B).load_state_dict(sd)
    v = torch.zeros((B, 6), dtype=torch.float64, device=dev)
    st = torch.zeros(B, dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        for i in range(20):
            eng.compute_velocity_dev(I_cur, I_des, Z, K, _lib.SELECT_ORDER, orders[i % 32], None, False, v, st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            eng.compute_velocity_dev(I_cur, I_des, Z, K, _lib.SELECT_ORDER, orders[i % 32], None, False, v, st)
        torch.cuda.synchronize()
        one = (time.perf_counter() - t0) / n
    print(f"{args.config} {args.precision} pairs {B}: one in flight {B / one:.1f} updates/s ({one * 1e3:.4f} ms), v_c {v[0].cpu().numpy()}")
    # per-class kernel times of instrumented updates
    lib = eng.lib
    lib.vitvs_timing_enable(eng.handle, 1)
    reps = 10
    with torch.cuda.stream(stream):
        for i in range(reps):
            eng.compute_velocity_dev(I_cur, I_des, Z, K, _lib.SELECT_ORDER, orders[i % 32], None, False, v, st)
        torch.cuda.synchronize()
    ncls = lib.vitvs_timing_classes()
    tot = (C.c_double * ncls)()
    cnt = (C.c_int32 * ncls)()
    lib.vitvs_timing_collect(eng.handle, ncls, tot, cnt)
    lib.vitvs_timing_enable(eng.handle, 0)
    total_us = 0.0
    for c in range(ncls):
        if cnt[c]:
            name = lib.vitvs_timing_class_name(c).decode()
            per_update = tot[c] * 1e3 / reps
            total_us += per_update
            print(f"  {name:12s} {cnt[c] // reps:3d} launches/update  {tot[c] * 1e3 / cnt[c]:7.2f} us each  {per_update:8.1f} us/update")
    print(f"  sum of kernel durations {total_us:.1f} us/update")
    eng.close()
    if args.depth > 1:
        pipe = UpdatePipeline(cfg, params, sd, precision=args.precision, depth=args.depth, max_pairs=B, device=dev, plan_hint=not args.no_hint)
        for i in range(30):
            pipe.submit(I_cur, I_des, Z, K, _lib.SELECT_ORDER, orders[i % 32], None, False)
        pipe.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            pipe.submit(I_cur, I_des, Z, K, _lib.SELECT_ORDER, orders[i % 32], None, False)
        pipe.synchronize()
        el = (time.perf_counter() - t0) / n
        print(f"{args.config} {args.precision} pairs {B}: {args.depth} in flight {B / el:.1f} updates/s ({el * 1e3:.4f} ms per update)")
        pipe.close()


if __name__ == "__main__":
    main()
