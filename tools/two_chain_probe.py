"""Would ONE update be shorter with its two forwards (goal frame, current frame) on two queues?  Times, for the descriptor forward alone
(vitvs_extract_descriptors_dev, bf16, ViT-B/16 224²): both frames through one handle on one stream, against one frame through each of
two handles (shared weights) on two high-priority streams at the same time, joined by the host.  The rest of an update (Gram + law,
~15 us) is common to both forms.

  python tools/two_chain_probe.py [--precision bf16] [--config vitb16_224] [--reps 300]
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
from vitvs_amd import config, synth, weights  # noqa: E402
from vitvs_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--config", default="vitb16_224")
    ap.add_argument("--reps", type=int, default=300)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = config.baseline_config(a.config)
    params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=False)
    sd = weights.synthetic_state_dict(cfg, 0)
    des, cur = synth.frame_pair(cfg.img_size, synth.RIG8_FRAME_SEEDS[0] if a.config == "vitb16_224" else synth.ACCEPTED_FRAME_SEEDS[a.config])
    both = torch.from_numpy(np.stack([des, cur])).to(dev)
    f_des, f_cur = both[0:1].contiguous(), both[1:2].contiguous()
    e2 = Engine(cfg, params, precision=a.precision, max_pairs=1).load_state_dict(sd)
    ea = Engine(cfg, params, precision=a.precision, max_pairs=1).share_weights(e2)
    eb = Engine(cfg, params, precision=a.precision, max_pairs=1).share_weights(e2)
    out2 = torch.empty((2, 1, cfg.tokens, cfg.dim), dtype=torch.float32, device=dev)
    outa = torch.empty((1, 1, cfg.tokens, cfg.dim), dtype=torch.float32, device=dev)
    outb = torch.empty((1, 1, cfg.tokens, cfg.dim), dtype=torch.float32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    s0 = torch.cuda.Stream(device=dev, priority=-1)
    s1 = torch.cuda.Stream(device=dev, priority=-1)
    sp = lambda s: C.c_void_p(s.cuda_stream)   # noqa: E731
    lib = e2.lib

    def one_stream():
        assert lib.vitvs_extract_descriptors_dev(e2.handle, 2, p(both), p(out2), sp(s0)) == 0
        s0.synchronize()

    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(2)                 # one host thread per chain: a chain is 84 launch calls (ctypes releases the GIL in them)

    def chain(handle, f, out, s):
        rc = lib.vitvs_extract_descriptors_dev(handle, 1, p(f), p(out), sp(s))
        s.synchronize()
        return rc

    def two_streams():
        fa = pool.submit(chain, ea.handle, f_des, outa, s0)
        fb = pool.submit(chain, eb.handle, f_cur, outb, s1)
        assert fa.result() == 0 and fb.result() == 0

    def one_frame():
        assert lib.vitvs_extract_descriptors_dev(ea.handle, 1, p(f_des), p(outa), sp(s0)) == 0
        s0.synchronize()

    for name, fn in (("both frames, one handle, one stream", one_stream), ("one frame per handle, two streams at once", two_streams),
                     ("one frame alone", one_frame)):
        for _ in range(20):
            fn()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            fn()
        print(f"{name:45s} {(time.perf_counter() - t0) / a.reps * 1e6:8.1f} us per forward pair (host-synchronised)")
    same = torch.equal(out2[0], outa[0]) and torch.equal(out2[1], outb[0])
    print("descriptors bit-identical between the two forms:", same, "(a 197-row launch takes another tile / slice plan than a 394-row one)",
          "max |diff|", float((out2[0] - outa[0]).abs().max()))


if __name__ == "__main__":
    main()
