#!/usr/bin/env python3
"""How closely do the fp32 / f16x2 (split-f16) / fp16 / bf16 modes follow the fp32 CPU oracle on YOUR checkpoint?

No pretrained weights can be fetched in the build environment, so the repository's own evidence about trained checkpoints is a
stand-in (weights.trained_like_state_dict).  With a local DINO / DINOv2 / timm state dict this prints the same
table for the real thing:

    python tests/check_checkpoint.py --model dinov2_vits14 --size 308 --weights /path/to/dinov2_vits14.pth [--binned] [--frames a.png b.png]

`--weights` is anything torch.load / safetensors can read into {name: tensor} with the reference's key names (patch_embed.proj.*,
cls_token, pos_embed, blocks.N.*; a DINOv2 hub checkpoint or a timm ViT state dict, dinov2_extractor.py:65-83).  Without `--frames`
a synthetic textured pair is used (vit-vs_amd/synth.py).  The oracle (oracle/) is test infrastructure, which is why this script lives
under tests/ (it is not collected by pytest): it runs the oracle as the checker, never as the thing measured or shipped.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (lives under tests/: only tests/, smoke() and bench.py's cpu_baseline may run the oracle)
sys.path.insert(0, ROOT)
import vitvs_amd  # noqa: E402,F401
from vitvs_amd import _lib, config, synth, weights  # noqa: E402
from vitvs_amd.engine import Engine  # noqa: E402
from oracle import servo_ref as sr  # noqa: E402
from oracle import vit_ref  # noqa: E402


def load_state_dict(path):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    sd = torch.load(path, map_location="cpu")
    for key in ("state_dict", "model", "teacher"):
        if isinstance(sd, dict) and key in sd and isinstance(sd[key], dict):
            sd = sd[key]
    return {k.replace("module.", "").replace("backbone.", ""): v.float() for k, v in sd.items() if torch.is_tensor(v)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="dinov2_vits14")
    ap.add_argument("--size", type=int, default=308)
    ap.add_argument("--stride", type=int, default=None)
    ap.add_argument("--weights", default=None, help="state dict file; default: the trained-like synthetic stand-in")
    ap.add_argument("--binned", action="store_true")
    ap.add_argument("--frames", nargs=2, default=None, help="goal and current image files (any size; resized like the reference)")
    args = ap.parse_args()
    cfg = config.vit_config(args.model, args.size, stride=args.stride) if args.stride else config.vit_config(args.model, args.size)
    sd = load_state_dict(args.weights) if args.weights else weights.trained_like_state_dict(cfg, 3)
    if args.frames:
        from PIL import Image
        des, cur = (np.asarray(Image.open(f).convert("RGB").resize((cfg.img_size, cfg.img_size)), dtype=np.uint8) for f in args.frames)
    else:
        des, cur = synth.frame_pair(cfg.img_size, 20250715)
    params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=args.binned)
    toks = vit_ref.block_tokens(sd, np.stack([des, cur]), patch=cfg.patch, stride=cfg.stride, heads=cfg.heads, layer=cfg.layer,
                                mean=cfg.mean, std=cfg.std)[:, 1:]
    if args.binned:
        toks = vit_ref.log_bin(toks, cfg.grid)
    S = sr.cosine_matrix(toks[0], toks[1], exact_order=False).numpy()
    n1r, n2r = S.argmax(1), S.argmax(0)
    top2 = np.sort(S, axis=1)[:, -2:]
    t = cfg.tokens
    print(f"{args.model} {cfg.img_size}x{cfg.img_size}, {t} tokens{' (binned)' if args.binned else ''}: oracle mutual NNs "
          f"{int((n2r[n1r] == np.arange(t)).sum())}, mean sim_1 {S.max(1).mean():.4f}, top-1 / top-2 margin median "
          f"{np.median(top2[:, 1] - top2[:, 0]):.2e} min {(top2[:, 1] - top2[:, 0]).min():.2e}")
    depth = synth.depth_pattern()
    order = np.random.default_rng(1).permutation(t).astype(np.int32)
    for precision in ("fp32", "f16x2", "fp16", "bf16"):
        eng = Engine(cfg, params, precision=precision, max_pairs=1).load_state_dict(sd)
        v, st = eng.compute_velocity(cur, des, depth, params.intrinsics(), mode=_lib.SELECT_ORDER, selection=order[None])
        det = eng.last_details(1)
        d1, d2 = det["nn_1"][0].astype(np.int64), det["nn_2"][0].astype(np.int64)
        d = eng.extract_descriptors(np.stack([des, cur])).double().cpu()[:, 0]
        dn = d / d.norm(dim=-1, keepdim=True).clamp_min(1e-8)
        sim_err = float(np.abs((dn[0] @ dn[1].T).numpy() - S).max())
        gap = max((float(S[i, n1r[i]] - S[i, d1[i]]) for i in np.nonzero(d1 != n1r)[0]), default=0.0)
        print(f"  {precision}: status {int(st[0])}, max |S_device - S_oracle| {sim_err:.2e}, arg-max agreement nn_1 {(d1 == n1r).mean():.4f} "
              f"nn_2 {(d2 == n2r).mean():.4f}, largest oracle gap of a differing nn_1 {gap:.2e}, finite tokens {bool(torch.isfinite(d).all())}")
        eng.close()


if __name__ == "__main__":
    main()
