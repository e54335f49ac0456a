"""Extractor weights: state-dict layout, seeded synthetic checkpoints, pos-embed resampling.

Tensor names follow the DINO / timm / DINOv2 state dicts the reference loads
(reference: dinov2_extractor.py:65-83; SURVEY.md §8(c) "weight names to accept").
There are no pretrained checkpoints offline, so benchmarks and tests use
``synthetic_state_dict`` (SURVEY.md §8(d) recipe: trunc-normal(0.02) matrices,
N(0, 0.02²) cls/pos; biases and norm/LayerScale gains are drawn non-trivially so
that every affine term of the forward is exercised by the parity tests).
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from .config import ViTConfig


def expected_tensors(cfg: ViTConfig) -> Dict[str, tuple]:
    """name -> shape for every tensor the forward (through block ``cfg.layer``) reads."""
    d, h, p = cfg.dim, cfg.hidden, cfg.patch
    g = cfg.native_grid
    out = {
        "patch_embed.proj.weight": (d, 3, p, p),
        "patch_embed.proj.bias": (d,),
        "cls_token": (1, 1, d),
        "pos_embed": (1, 1 + g * g, d),
    }
    for i in range(cfg.blocks_run):
        b = f"blocks.{i}."
        out[b + "norm1.weight"] = (d,)
        out[b + "norm1.bias"] = (d,)
        out[b + "attn.qkv.weight"] = (3 * d, d)
        out[b + "attn.qkv.bias"] = (3 * d,)
        out[b + "attn.proj.weight"] = (d, d)
        out[b + "attn.proj.bias"] = (d,)
        out[b + "norm2.weight"] = (d,)
        out[b + "norm2.bias"] = (d,)
        out[b + "mlp.fc1.weight"] = (h, d)
        out[b + "mlp.fc1.bias"] = (h,)
        out[b + "mlp.fc2.weight"] = (d, h)
        out[b + "mlp.fc2.bias"] = (d,)
        if cfg.layerscale:
            out[b + "ls1.gamma"] = (d,)
            out[b + "ls2.gamma"] = (d,)
    return out


def synthetic_state_dict(cfg: ViTConfig, seed: int = 0, affine_jitter: bool = True) -> Dict[str, torch.Tensor]:
    """Deterministic fp32 CPU state dict for ``cfg`` (same bytes on every machine for a seed)."""
    gen = torch.Generator(device="cpu")
    gen.manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in expected_tensors(cfg).items():
        leaf = name.rsplit(".", 1)[-1]
        if name in ("cls_token", "pos_embed"):
            t = torch.randn(shape, generator=gen) * 0.02
        elif leaf == "gamma":
            t = 1.0 + (torch.rand(shape, generator=gen) - 0.5) * (0.5 if affine_jitter else 0.0)
        elif "norm" in name and leaf == "weight":
            t = 1.0 + torch.randn(shape, generator=gen) * (0.05 if affine_jitter else 0.0)
        elif leaf == "bias":
            t = torch.randn(shape, generator=gen) * (0.02 if affine_jitter else 0.0)
        else:  # Linear / conv matrices: trunc-normal(std 0.02, +-2 std)
            t = torch.empty(shape)
            torch.nn.init.trunc_normal_(t, mean=0.0, std=0.02, a=-0.04, b=0.04, generator=gen)
        sd[name] = t.to(torch.float32).contiguous()
    return sd


def trained_like_state_dict(cfg: ViTConfig, seed: int = 0, qk_gain: float = 9.0, outlier_channels: int = 4,
                            outlier_gain: float = 150.0) -> Dict[str, torch.Tensor]:
    """A synthetic checkpoint with the statistics a TRAINED ViT shows and trunc-normal(0.02) initialisation does not
    (no checkpoints offline): the query / key rows of every ``attn.qkv.weight`` are scaled by ``qk_gain`` so that the
    attention logits spread over several units (peaky softmax rows, entropy of 1-2.5 nats at 197 tokens instead of the
    near-uniform ln 197 = 5.3), and ``outlier_channels`` residual channels end up 40-60 x larger than the median channel
    (the "massive activation" channels of trained ViTs: the rows of the first two blocks' ``mlp.fc2.weight`` / bias that
    write those channels are scaled), which dominate every LayerNorm's statistics from block 1 on and stress the 16-bit
    GEMMs' dynamic range.  Deterministic for a seed; used by the stress parity tests only."""
    sd = synthetic_state_dict(cfg, seed)
    d = cfg.dim
    gen = torch.Generator(device="cpu")
    gen.manual_seed(seed + 977)
    chans = torch.randperm(d, generator=gen)[:outlier_channels]
    for i in range(cfg.blocks_run):
        w = sd[f"blocks.{i}.attn.qkv.weight"]
        w[: 2 * d] *= qk_gain                                  # q and k rows (the bias is scaled with them)
        sd[f"blocks.{i}.attn.qkv.bias"][: 2 * d] *= qk_gain
        if i < 2:
            sd[f"blocks.{i}.mlp.fc2.weight"][chans] *= outlier_gain
            sd[f"blocks.{i}.mlp.fc2.bias"][chans] = outlier_gain * 0.02 * torch.sign(sd[f"blocks.{i}.mlp.fc2.bias"][chans] + 1e-12)
    return sd


def resample_pos_embed(pos_embed: torch.Tensor, grid: int) -> torch.Tensor:
    """Positional encoding for a ``grid x grid`` token grid, shape (1 + grid², D), fp32.

    Bicubic resampling of the stored square grid with the "+0.1" scale-factor
    trick (reference: dinov2_extractor.py:94-118; DINO / DINOv2 use the same
    formula when stride == patch).  Input-size-only, so it runs once at load
    time on the host.
    """
    pos_embed = pos_embed.to(torch.float32)
    n_stored = pos_embed.shape[1] - 1
    side = int(math.sqrt(n_stored))
    if side * side != n_stored:
        raise ValueError("stored pos_embed is not a square grid")
    if side == grid:
        return pos_embed[0].contiguous()
    dim = pos_embed.shape[-1]
    cls_pos = pos_embed[:, 0]
    patch_pos = pos_embed[:, 1:].reshape(1, side, side, dim).permute(0, 3, 1, 2)
    scale = (grid + 0.1) / side
    patch_pos = torch.nn.functional.interpolate(
        patch_pos, scale_factor=(scale, scale), mode="bicubic", align_corners=False,
        recompute_scale_factor=False)
    if patch_pos.shape[-1] != grid or patch_pos.shape[-2] != grid:
        raise RuntimeError(f"pos-embed resample produced {tuple(patch_pos.shape[-2:])}, wanted {grid}")
    patch_pos = patch_pos.permute(0, 2, 3, 1).reshape(1, grid * grid, dim)
    return torch.cat((cls_pos.unsqueeze(0), patch_pos), dim=1)[0].contiguous()


def check_state_dict(cfg: ViTConfig, sd: Dict[str, torch.Tensor]) -> None:
    want = expected_tensors(cfg)
    missing = [k for k in want if k not in sd]
    if missing:
        raise KeyError(f"state dict lacks {len(missing)} tensors, e.g. {missing[:4]}")
    for k, shape in want.items():
        if tuple(sd[k].shape) != tuple(shape):
            raise ValueError(f"{k}: shape {tuple(sd[k].shape)} != expected {shape}")
