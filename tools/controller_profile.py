"""cProfile of the drop-in Controller's step (servo.Controller.ibvs on 640 x 480 host frames + depth, bf16) — where the host time of
`controller_loop` goes beside the C calls.

  python tools/controller_profile.py [order|reference] [updates]
"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vitvs_amd  # noqa: E402,F401
from vitvs_amd import config, synth, weights, servo  # noqa: E402
from vitvs_amd.engine import Engine  # noqa: E402


def main():
    sel = sys.argv[1] if len(sys.argv) > 1 else "order"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    cfg = config.baseline_config("vitb16_224")
    params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=False)
    sd = weights.synthetic_state_dict(cfg, 0)
    des, cur = synth.frame_pair(cfg.img_size, synth.RIG8_FRAME_SEEDS[0])
    cam = lambda a: np.asarray(Image.fromarray(a).resize((params.u_max, params.v_max)), dtype=np.uint8)   # noqa: E731
    goal_cam, cur_cam = cam(des), cam(cur)
    depth = synth.depth_pattern()
    eng = Engine(cfg, params, precision="bf16", max_pairs=1).load_state_dict(sd)
    ctl = servo.Controller(eng, goal_image=goal_cam, selection=sel)
    ctl.generator = torch.Generator().manual_seed(121)

    def step():
        ctl.image_callback_rgb(cur_cam)
        ctl.image_callback_depth(depth)
        ctl.ibvs()
    for _ in range(40):
        step()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    print(f"selection {sel}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms per step (callbacks + ibvs)")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        step()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
