"""Which operand rounding breaks the arg-max on trained-like weights?  (CPU, oracle-side)

The forward of oracle/vit_ref.py restated with a rounding hook on every matrix operand family, run on
``weights.trained_like_state_dict`` with ONE family rounded at a time (and with all of them), in three roundings:

  fp16  : 11-bit significand (the fp16 mode's operands)
  bf16  : 8-bit significand (the headline dtype's operands)
  f16x2 : hi + lo, two fp16 numbers (22 significant bits; the split-f16 precision of csrc/gemm_core.h — the contraction
          hi.hi + hi.lo + lo.hi in fp32, the lo.lo term dropped; weights pre-scaled by a power of two)

For each: max |S - S_oracle| of the cosine matrix and the arg-max agreement of both tables with the unrounded fp32 oracle.
Test infrastructure (imports oracle/): never part of the product.

  python tests/precision_ablation.py [vitb16_224] [vitl14_518]  > profiles/r05_precision_ablation.txt
"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vitvs_amd  # noqa: E402,F401
from vitvs_amd import config, synth, weights  # noqa: E402
from oracle import servo_ref as sr  # noqa: E402
from oracle import vit_ref  # noqa: E402

FAMILIES = ["weights", "patches->embed", "xn->qkv", "q,k->QK^T", "P,V->PV", "attn->proj", "xn->fc1", "hid->fc2"]


def r_fp16(t):
    return t.half().float()


def r_bf16(t):
    return t.bfloat16().float()


def split2(t, scale_pow2=0):
    s = 2.0 ** scale_pow2
    ts = t * s
    hi = ts.half().float().clamp(-65504, 65504)
    lo = (ts - hi).half().float()
    return hi, lo, s


def r_x2(t):
    hi, lo, s = split2(t)
    return (hi + lo) / s


def weight_pow2(w):
    m = float(w.abs().max())
    return 0 if m == 0 else int(math.floor(14 - math.log2(m) - 1e-9)) - 1   # max |w| 2^s in [2^12, 2^13]


class Rounder:
    """mm(family_a, a, family_b, b): a @ b^T with the operand families in `on` rounded by `kind`.  f16x2 with both operands
    rounded is the three-term contraction in fp32 (what the kernel computes); one-sided it is (hi + lo) of that side."""

    def __init__(self, kind, on):
        self.kind, self.on = kind, set(on)

    def r(self, fam, t, is_weight=False):
        if fam not in self.on:
            return t
        if self.kind == "fp16":
            return r_fp16(t)
        if self.kind == "bf16":
            return r_bf16(t)
        if is_weight:
            hi, lo, s = split2(t, weight_pow2(t))
            return (hi + lo) / s
        return r_x2(t)

    def mm(self, fa, a, fb, b, b_is_weight=False):
        if self.kind == "f16x2" and fa in self.on and fb in self.on:
            ah, al, sa = split2(a)
            bh, bl, sb = split2(b, weight_pow2(b) if b_is_weight else 0)
            out = ah @ bl.transpose(-2, -1) + al @ bh.transpose(-2, -1)
            return (out + ah @ bh.transpose(-2, -1)) / (sa * sb)
        return self.r(fa, a) @ self.r(fb, b, b_is_weight).transpose(-2, -1)


@torch.no_grad()
def forward(sd, frames, cfg, rd):
    x = vit_ref.preprocess_u8(frames, cfg.mean, cfg.std)
    p, st = cfg.patch, cfg.stride
    cols = F.unfold(x, kernel_size=p, stride=st).transpose(1, 2)                         # [B, T, 3 p p]
    w = sd["patch_embed.proj.weight"].reshape(cfg.dim, -1)
    x = rd.mm("patches->embed", cols, "weights", w, True) + sd["patch_embed.proj.bias"]
    b = x.shape[0]
    g = int(math.isqrt(x.shape[1]))
    x = torch.cat((sd["cls_token"].expand(b, -1, -1), x), dim=1) + vit_ref.resample_pos_embed(sd["pos_embed"], g)
    d, H = cfg.dim, cfg.heads
    for i in range(cfg.layer + 1):
        pfx = f"blocks.{i}."
        y = F.layer_norm(x, (d,), sd[pfx + "norm1.weight"], sd[pfx + "norm1.bias"], 1e-6)
        qkv = rd.mm("xn->qkv", y, "weights", sd[pfx + "attn.qkv.weight"], True) + sd[pfx + "attn.qkv.bias"]
        n = qkv.shape[1]
        q, k, v = (t.transpose(1, 2) for t in qkv.reshape(b, n, 3, H, d // H).unbind(2))
        s = rd.mm("q,k->QK^T", q, "q,k->QK^T", k) * (d // H) ** -0.5
        pr = s.softmax(dim=-1)
        o = rd.mm("P,V->PV", pr, "P,V->PV", v.transpose(-2, -1)).transpose(1, 2).reshape(b, n, d)
        y = rd.mm("attn->proj", o, "weights", sd[pfx + "attn.proj.weight"], True) + sd[pfx + "attn.proj.bias"]
        if pfx + "ls1.gamma" in sd:
            y = y * sd[pfx + "ls1.gamma"]
        x = x + y
        y = F.layer_norm(x, (d,), sd[pfx + "norm2.weight"], sd[pfx + "norm2.bias"], 1e-6)
        y = F.gelu(rd.mm("xn->fc1", y, "weights", sd[pfx + "mlp.fc1.weight"], True) + sd[pfx + "mlp.fc1.bias"])
        y = rd.mm("hid->fc2", y, "weights", sd[pfx + "mlp.fc2.weight"], True) + sd[pfx + "mlp.fc2.bias"]
        if pfx + "ls2.gamma" in sd:
            y = y * sd[pfx + "ls2.gamma"]
        x = x + y
    return x


def similarity(tokens):
    return sr.cosine_matrix(tokens[0, 1:], tokens[1, 1:], exact_order=False).numpy()


def main():
    keys = sys.argv[1:] or ["vitb16_224", "vitl14_518"]
    torch.set_num_threads(os.cpu_count() or 1)
    for key in keys:
        cfg = config.baseline_config(key)
        sd = weights.trained_like_state_dict(cfg, 3)                                      # the weights of test_trained_like_statistics_end_to_end
        seed = synth.RIG8_FRAME_SEEDS[0] if key == "vitb16_224" else synth.ACCEPTED_FRAME_SEEDS[key]
        des, cur = synth.frame_pair(cfg.img_size, seed)
        frames = np.stack([des, cur])
        ref_tokens = vit_ref.block_tokens(sd, frames, patch=cfg.patch, stride=cfg.stride, heads=cfg.heads, layer=cfg.layer,
                                          mean=cfg.mean, std=cfg.std)
        S = similarity(ref_tokens)
        check = forward(sd, frames, cfg, Rounder("fp16", []))
        scale = float(ref_tokens.abs().max())
        n1, n2 = S.argmax(1), S.argmax(0)
        top2 = np.sort(S, axis=1)[:, -2:]
        print(f"== {key}: {cfg.tokens} tokens, trained-like weights (seed 3); oracle top-1/top-2 margin median "
              f"{np.median(top2[:, 1] - top2[:, 0]):.2e}; hooked forward vs oracle/vit_ref.py with nothing rounded: "
              f"max |dtoken| / scale {float((check - ref_tokens).abs().max()) / scale:.1e}")
        print(f"{'rounding':6s} {'family':16s} {'max|dtoken|/scale':>18s} {'max|S-S_oracle|':>16s} {'nn_1 agree':>11s} {'nn_2 agree':>11s} {'worst oracle gap':>17s}")
        for kind in ("bf16", "fp16", "f16x2"):
            for fams in [[f] for f in FAMILIES] + [FAMILIES[1:], FAMILIES]:
                name = fams[0] if len(fams) == 1 else ("all activations" if len(fams) == len(FAMILIES) - 1 else "ALL")
                tok = forward(sd, frames, cfg, Rounder(kind, fams))
                Sd = similarity(tok)
                d1, d2 = Sd.argmax(1), Sd.argmax(0)
                gap = 0.0
                for got, ref, M in ((d1, n1, S), (d2, n2, S.T)):
                    bad = np.nonzero(got != ref)[0]
                    gap = max([gap] + [float(M[i, ref[i]] - M[i, got[i]]) for i in bad])
                print(f"{kind:6s} {name:16s} {float((tok - ref_tokens).abs().max()) / scale:18.2e} {float(np.abs(Sd - S).max()):16.2e} "
                      f"{float((d1 == n1).mean()):11.4f} {float((d2 == n2).mean()):11.4f} {gap:17.2e}")
        print()


if __name__ == "__main__":
    main()
