"""ORACLE (test infrastructure, never shipped): fp32 CPU restatement of the ViT forward
the reference runs through ``ViTExtractor`` for the descriptor it servoes on.

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may import
this file.  The product path (vit-vs_amd/) never does.

What it restates, and from where (paths under /root/reference):
  * preprocessing ``ToTensor`` + ``Normalize``        catkin_ws/ibvs/src/dinov2_extractor.py:177-191, 49-50
  * patch embed (Conv2d k=p, stride), cls, pos_embed   dinov2_extractor.py:85-144 (stride hack, pos-enc resample)
  * ``Block.forward`` (inference branch)               dino_patch/block.py:90-96, 112-115
  * ``Attention.forward``                              dino_patch/attention.py:70-80 (scale hd^-0.5 at :51)
  * which tensor is the descriptor                     dinov2_extractor.py:193-263, 313-337 (output of blocks[11],
                                                       before the final norm, cls dropped)
The ViT arithmetic itself lives in third-party code that is NOT under /root/reference
(timm==0.6.12, facebookresearch/dino@main, facebookresearch/dinov2 — fetched by
torch.hub at run time, dinov2_extractor.py:65-83).  Their published definition is
restated here: pre-norm residual blocks, LayerNorm eps 1e-6, erf-GELU MLP with ratio 4,
qkv bias on, LayerScale (DINOv2 only).  PARITY UNPINNED at this boundary: the reference
holds no test, fixture or golden vector for the forward; the
restatement is cross-checked against dino_patch/attention.py and an independent
implementation (HF transformers ViTModel) in tests/test_oracle_vit.py.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F


def preprocess_u8(frames_u8: np.ndarray, mean, std) -> torch.Tensor:
    """uint8 [B,S,S,3] (RGB, HWC) -> float32 [B,3,S,S]; ToTensor then Normalize."""
    x = torch.from_numpy(np.ascontiguousarray(frames_u8)).permute(0, 3, 1, 2).to(torch.float32).div(255)
    m = torch.tensor(mean, dtype=torch.float32).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, 3, 1, 1)
    return (x - m) / s


def resample_pos_embed(pos_embed: torch.Tensor, grid: int) -> torch.Tensor:
    """(1, 1+G0², D) -> (1, 1+grid², D); dinov2_extractor.py:94-118."""
    n0 = pos_embed.shape[1] - 1
    side = int(math.sqrt(n0))
    if side == grid:
        return pos_embed
    dim = pos_embed.shape[-1]
    w0 = grid + 0.1
    patch = pos_embed[:, 1:].reshape(1, side, side, dim).permute(0, 3, 1, 2)
    patch = F.interpolate(patch, scale_factor=(w0 / side, w0 / side), mode="bicubic",
                          align_corners=False, recompute_scale_factor=False)
    assert patch.shape[-1] == grid and patch.shape[-2] == grid
    patch = patch.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((pos_embed[:, :1], patch), dim=1)


def attention(x: torch.Tensor, qkv_w, qkv_b, proj_w, proj_b, heads: int) -> torch.Tensor:
    """dino_patch/attention.py:70-80 with SDPA written out: softmax(q kᵀ · hd^-0.5) v."""
    b, n, c = x.shape
    hd = c // heads
    qkv = F.linear(x, qkv_w, qkv_b).reshape(b, n, 3, heads, hd)
    q, k, v = qkv.unbind(2)
    q, k, v = (t.transpose(1, 2) for t in (q, k, v))
    attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    attn = attn.softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(b, n, c)
    return F.linear(out, proj_w, proj_b)


def block(x: torch.Tensor, sd: Dict[str, torch.Tensor], i: int, heads: int, eps: float) -> torch.Tensor:
    """dino_patch/block.py:90-96,112-115: x + ls1(attn(norm1 x)); x + ls2(mlp(norm2 x))."""
    p = f"blocks.{i}."
    d = x.shape[-1]
    y = F.layer_norm(x, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
    y = attention(y, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"],
                  sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"], heads)
    if p + "ls1.gamma" in sd:
        y = y * sd[p + "ls1.gamma"]
    x = x + y
    y = F.layer_norm(x, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
    y = F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
    y = F.gelu(y)  # nn.GELU default = erf form (block.py:55)
    y = F.linear(y, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    if p + "ls2.gamma" in sd:
        y = y * sd[p + "ls2.gamma"]
    return x + y


@torch.no_grad()
def block_tokens(sd: Dict[str, torch.Tensor], frames_u8: np.ndarray, *, patch: int, stride: int, heads: int,
                 layer: int, mean, std, eps: float = 1e-6, return_all: bool = False):
    """Residual stream after ``blocks[layer]`` for uint8 frames [B,S,S,3] -> float32 [B, 1+T, D]."""
    x = preprocess_u8(frames_u8, mean, std)
    x = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=stride)
    b, d, gh, gw = x.shape
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat((sd["cls_token"].expand(b, -1, -1), x), dim=1)
    assert gh == gw, "square grids only (vitvs_v2.py:75)"
    x = x + resample_pos_embed(sd["pos_embed"], gh)
    stages = [x]
    for i in range(layer + 1):
        x = block(x, sd, i, heads, eps)
        if return_all:
            stages.append(x)
    return stages if return_all else x


@torch.no_grad()
def extract_descriptors(sd, frames_u8, *, patch, stride, heads, layer, mean, std, bin: bool = False,
                        eps: float = 1e-6) -> torch.Tensor:
    """``ViTExtractor.extract_descriptors(facet='token')``: [B,1,T,D] (or [B,1,T,9D] binned).
    dinov2_extractor.py:313-337; binning = :265-311 with hierarchy=1."""
    x = block_tokens(sd, frames_u8, patch=patch, stride=stride, heads=heads, layer=layer, mean=mean, std=std,
                     eps=eps)
    x = x[:, 1:, :]
    if bin:
        g = int(math.sqrt(x.shape[1]))
        x = log_bin(x, g)
    return x.unsqueeze(1)


@torch.no_grad()
def extract_facet(sd, frames_u8, *, patch, stride, heads, layer, mean, std, facet: str, eps: float = 1e-6,
                  bin: bool = False, include_cls: bool = False) -> torch.Tensor:
    """``ViTExtractor.extract_descriptors(facet='query'|'key'|'value'|'token', bin, include_cls)``: [B,1,T,D], [B,1,1+T,D]
    with ``include_cls`` or [B,1,T,9D] with ``bin`` (the two together are refused, dinov2_extractor.py:330-331).

    dinov2_extractor.py:193-217: the hook on ``blocks[layer].attn`` recomputes ``qkv = attn.qkv(norm1(x))`` reshaped to
    [3,B,H,N,hd] and keeps one of the three ([B,H,N,hd]); :326-334 drops the cls token and flattens with
    ``permute(0,2,3,1)``, i.e. descriptor index = d * H + h (head index fastest)."""
    assert not (bin and include_cls), "bin = True and include_cls = True are not supported together, set one of them False."
    if facet == "token":
        f = block_tokens(sd, frames_u8, patch=patch, stride=stride, heads=heads, layer=layer, mean=mean, std=std, eps=eps)
        f = f if include_cls else f[:, 1:]
        return (log_bin(f, int(math.sqrt(f.shape[1]))) if bin else f).unsqueeze(1)
    idx = {"query": 0, "key": 1, "value": 2}[facet]
    stages = block_tokens(sd, frames_u8, patch=patch, stride=stride, heads=heads, layer=layer, mean=mean, std=std,
                          eps=eps, return_all=True)
    x = stages[layer]                                        # input of blocks[layer]
    p = f"blocks.{layer}."
    d = x.shape[-1]
    y = F.layer_norm(x, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
    b, n, c = y.shape
    qkv = F.linear(y, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(b, n, 3, heads, c // heads)
    f = qkv.permute(2, 0, 3, 1, 4)[idx]                      # B x H x N x hd
    if not include_cls:
        f = f[:, :, 1:, :]
    f = f.permute(0, 2, 3, 1).flatten(start_dim=-2, end_dim=-1)     # B x t x (d x h)
    if bin:                                                  # _log_bin on the B x h x t x d tensor: the same 3x3 concatenation
        f = log_bin(f, int(math.sqrt(f.shape[1])))
    return f.unsqueeze(1)


def cls_attention(sd, frames_u8, *, patch, stride, heads, layer, mean, std, eps: float = 1e-6) -> torch.Tensor:
    """The 'attn' facet restricted to what the reference reads from it: attention probabilities of the class token in
    ``blocks[layer]`` over the patch tokens, [B, H, T] (dinov2_extractor.py:230-231 hooks ``attn.attn_drop``, i.e. the
    softmax output ``((q @ k^T) * hd^-0.5).softmax(-1)`` of shape B x H x N x N; :349 takes ``[:, heads, 0, 1:]``)."""
    stages = block_tokens(sd, frames_u8, patch=patch, stride=stride, heads=heads, layer=layer, mean=mean, std=std,
                          eps=eps, return_all=True)
    x = stages[layer]                                        # input of blocks[layer]
    p = f"blocks.{layer}."
    d = x.shape[-1]
    y = F.layer_norm(x, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
    b, n, c = y.shape
    qkv = F.linear(y, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(b, n, 3, heads, c // heads).permute(2, 0, 3, 1, 4)
    q, k = qkv[0], qkv[1]                                    # B x H x N x hd
    attn = ((q @ k.transpose(-2, -1)) * (c // heads) ** -0.5).softmax(dim=-1)
    return attn[:, :, 0, 1:]


def saliency_maps(sd, frames_u8, *, patch, stride, heads, layer, mean, std, head_idxs=(0, 2, 4, 5), eps: float = 1e-6):
    """``ViTExtractor.extract_saliency_maps`` (dinov2_extractor.py:339-353): mean of the chosen heads' class-token attention,
    min-max normalised per image, [B, T].  (The reference subtracts a [B] vector from a [B, T] map, which broadcasts as
    intended only for a batch of one — its only use; here every image is normalised by its own extremes.)"""
    a = cls_attention(sd, frames_u8, patch=patch, stride=stride, heads=heads, layer=layer, mean=mean, std=std, eps=eps)
    m = a[:, list(head_idxs)].mean(dim=1)
    lo, hi = m.min(dim=1)[0], m.max(dim=1)[0]
    return (m - lo[:, None]) / (hi - lo)[:, None]


def log_bin(tokens: torch.Tensor, grid: int) -> torch.Tensor:
    """hierarchy=1 log-binning: for each cell the 3x3 neighbourhood tokens concatenated in
    row-major (dy,dx) order, replicate-clamped at the border (dinov2_extractor.py:289-308).
    tokens [B,T,D] -> [B,T,9D]."""
    b, t, d = tokens.shape
    grid_t = tokens.reshape(b, grid, grid, d)
    idx = torch.arange(grid)
    parts = []
    for dy in (-1, 0, 1):
        iy = (idx + dy).clamp(0, grid - 1)
        for dx in (-1, 0, 1):
            ix = (idx + dx).clamp(0, grid - 1)
            parts.append(grid_t[:, iy][:, :, ix])
    return torch.cat(parts, dim=-1).reshape(b, t, 9 * d)
