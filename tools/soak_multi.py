"""Soak of servo.MultiController (tools only): N cameras through one UpdatePipeline (or one batched call per round) for many rounds
with a fixed set of frames and a re-seeded draw every `period` rounds — every camera's raw twist must repeat bit for bit with the
period, whichever slot computed it, and the device memory in use must not grow."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vitvs_amd  # noqa: F401
from vitvs_amd import config, servo, synth, weights
from vitvs_amd.engine import Engine
from vitvs_amd.pipeline import UpdatePipeline


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    backend = sys.argv[2] if len(sys.argv) > 2 else "pipeline"
    precision = sys.argv[3] if len(sys.argv) > 3 else "fp16"
    n_cam, period = 8, 5
    dev = torch.device("cuda", 0)
    cfg = config.baseline_config("vitb16_224")
    params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=False)
    sd = weights.synthetic_state_dict(cfg, 0)
    pairs = [synth.frame_pair(cfg.img_size, s) for s in synth.RIG8_FRAME_SEEDS[:n_cam]]
    be = (UpdatePipeline(cfg, params, sd, precision=precision, depth=3) if backend == "pipeline"
          else Engine(cfg, params, precision=precision, max_pairs=n_cam).load_state_dict(sd))
    mc = servo.MultiController(be, [p[0] for p in pairs])
    depth = synth.depth_pattern()
    free0 = None
    first, bad = {}, 0
    t0 = time.perf_counter()
    for r in range(rounds):
        if r % period == 0:
            torch.manual_seed(7)                         # the same draws every period
        for c in range(n_cam):
            mc.image_callback_rgb(c, np.roll(pairs[c][1], (r % period) - 2, axis=1))
            mc.image_callback_depth(c, depth)
        mc.ibvs()
        for c, cam in enumerate(mc.cameras):
            key, val = (r % period, c), np.asarray(cam._raw_v, np.float64).tobytes()
            bad += first.setdefault(key, val) != val
        if r == period:
            free0 = torch.cuda.mem_get_info(dev)[0]
    dt = time.perf_counter() - t0
    free1 = torch.cuda.mem_get_info(dev)[0]
    print(f"MultiController[{backend}, {precision}]: {rounds} rounds x {n_cam} cameras in {dt:.2f} s ({rounds * n_cam / dt:.0f} updates/s through the "
          f"host adapter), results differing from their first pass: {bad}, device memory in use changed by {(free0 - free1) / 2**20:.1f} MiB")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
