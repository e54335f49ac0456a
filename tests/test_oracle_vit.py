"""Cross-checks of the oracle's ViT restatement (oracle/vit_ref.py): against the reference's
vendored DINOv2 attention (dino_patch/attention.py, only where /root/reference is mounted)
and against an independent implementation (HF transformers ViTModel / Dinov2Model built from
local config objects — no download).  These do not pin the reference's third-party forward
(parity unpinned, see oracle/vit_ref.py); they guard the restatement against slips."""
import dataclasses

import numpy as np
import pytest
import torch

import vitvs_amd  # noqa: F401
from vitvs_amd import config, weights
from oracle import ref_extract as rx
from oracle import vit_ref


def _tiny_cfg(layerscale: bool):
    base = config.vit_config("dinov2_vits14" if layerscale else "dino_vits16", 56 if layerscale else 64)
    return dataclasses.replace(base, dim=128, depth=3, heads=2, layer=2, native_grid=base.grid)


def _frames(cfg, n=2, seed=0):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(n, cfg.img_size, cfg.img_size, 3), dtype=np.uint8)


def _oracle_tokens(cfg, sd, frames, return_all=False):
    return vit_ref.block_tokens(sd, frames, patch=cfg.patch, stride=cfg.stride, heads=cfg.heads, layer=cfg.layer,
                                mean=cfg.mean, std=cfg.std, return_all=return_all)


@pytest.mark.skipif(not rx.available(), reason="reference tree not mounted")
def test_attention_matches_vendored_dinov2_attention():
    mod = rx.load_attention_module()
    torch.manual_seed(0)
    att = mod.Attention(128, num_heads=2, qkv_bias=True).eval()
    x = torch.randn(2, 17, 128)
    with torch.no_grad():
        ref = att(x)
        mine = vit_ref.attention(x, att.qkv.weight, att.qkv.bias, att.proj.weight, att.proj.bias, 2)
    assert float((ref - mine).abs().max()) < 5e-6


def _hf_vit_hidden_states(cfg, sd, frames):
    """HF ViTModel built from a local config with the oracle's weights: (model, preprocessed input, hidden states)."""
    import transformers
    hf_cfg = transformers.ViTConfig(hidden_size=cfg.dim, num_hidden_layers=cfg.blocks_run,
                                    num_attention_heads=cfg.heads, intermediate_size=cfg.hidden,
                                    image_size=cfg.img_size, patch_size=cfg.patch, layer_norm_eps=cfg.ln_eps,
                                    hidden_act="gelu", qkv_bias=True, hidden_dropout_prob=0.0,
                                    attention_probs_dropout_prob=0.0)
    model = transformers.ViTModel(hf_cfg, add_pooling_layer=False).eval()
    m = {"embeddings.cls_token": sd["cls_token"], "embeddings.position_embeddings": sd["pos_embed"],
         "embeddings.patch_embeddings.projection.weight": sd["patch_embed.proj.weight"],
         "embeddings.patch_embeddings.projection.bias": sd["patch_embed.proj.bias"]}
    d = cfg.dim
    for i in range(cfg.blocks_run):
        s, t = f"blocks.{i}.", f"layers.{i}."
        qw, qb = sd[s + "attn.qkv.weight"], sd[s + "attn.qkv.bias"]
        for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
            m[t + f"attention.{nm}.weight"] = qw[j * d:(j + 1) * d]
            m[t + f"attention.{nm}.bias"] = qb[j * d:(j + 1) * d]
        m[t + "attention.o_proj.weight"] = sd[s + "attn.proj.weight"]
        m[t + "attention.o_proj.bias"] = sd[s + "attn.proj.bias"]
        m[t + "layernorm_before.weight"] = sd[s + "norm1.weight"]
        m[t + "layernorm_before.bias"] = sd[s + "norm1.bias"]
        m[t + "layernorm_after.weight"] = sd[s + "norm2.weight"]
        m[t + "layernorm_after.bias"] = sd[s + "norm2.bias"]
        m[t + "mlp.fc1.weight"] = sd[s + "mlp.fc1.weight"]
        m[t + "mlp.fc1.bias"] = sd[s + "mlp.fc1.bias"]
        m[t + "mlp.fc2.weight"] = sd[s + "mlp.fc2.weight"]
        m[t + "mlp.fc2.bias"] = sd[s + "mlp.fc2.bias"]
    missing, unexpected = model.load_state_dict(m, strict=False)
    assert not [k for k in missing if not k.startswith("layernorm.")], missing
    assert not unexpected
    x = vit_ref.preprocess_u8(frames, cfg.mean, cfg.std)
    with torch.no_grad():
        hs = model(pixel_values=x, output_hidden_states=True).hidden_states
    return model, x, hs


def test_forward_matches_hf_vit():
    pytest.importorskip("transformers")
    cfg = _tiny_cfg(False)
    sd = weights.synthetic_state_dict(cfg, 1)
    frames = _frames(cfg)
    model, x, hs = _hf_vit_hidden_states(cfg, sd, frames)
    mine = _oracle_tokens(cfg, sd, frames, return_all=True)
    assert len(hs) == len(mine)
    for a, b in zip(hs, mine):
        assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max()))
    # the 'attn' facet / saliency maps (dinov2_extractor.py:230-231, 339-353): the class token's attention row of the last
    # block against the independent implementation's attention probabilities
    model.config._attn_implementation = "eager"
    with torch.no_grad():
        att = model(pixel_values=x, output_attentions=True).attentions
    if att is not None and att[cfg.layer] is not None:
        kw = dict(patch=cfg.patch, stride=cfg.stride, heads=cfg.heads, layer=cfg.layer, mean=cfg.mean, std=cfg.std)
        cls_rows = vit_ref.cls_attention(sd, frames, **kw)
        assert cls_rows.shape == (2, cfg.heads, cfg.tokens)
        assert float((cls_rows - att[cfg.layer][:, :, 0, 1:]).abs().max()) < 1e-6
        sal = vit_ref.saliency_maps(sd, frames, head_idxs=(0, 1), **kw)
        assert sal.shape == (2, cfg.tokens) and float(sal.min()) == 0.0 and float(sal.max()) == 1.0
        want = att[cfg.layer][:, :, 0, 1:].mean(dim=1)
        want = (want - want.min(dim=1)[0][:, None]) / (want.max(dim=1)[0] - want.min(dim=1)[0])[:, None]
        assert float((sal - want).abs().max()) < 1e-4


def test_forward_matches_hf_vit_at_vitb16_width():
    """The same witness at the headline model's dimensions (ViT-B/16 224²: 768 wide, 12 heads, 3072 hidden, 197 tokens; two
    blocks suffice — every block is the same arithmetic): patch embedding, cls / pos-embed, attention and MLP at full width."""
    pytest.importorskip("transformers")
    cfg = dataclasses.replace(config.baseline_config("vitb16_224"), depth=2, layer=1)
    sd = weights.synthetic_state_dict(cfg, 4)
    frames = _frames(cfg, n=2, seed=3)
    _, _, hs = _hf_vit_hidden_states(cfg, sd, frames)
    mine = _oracle_tokens(cfg, sd, frames, return_all=True)
    assert len(hs) == len(mine) == 3
    for a, b in zip(hs, mine):
        assert a.shape == b.shape == (2, 197, 768)
        assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max()))


def _hf_dinov2_hidden_states(cfg, sd, frames):
    """HF Dinov2Model from a local config with the oracle's weights.  The position embedding handed over is the oracle's
    RESAMPLED one (HF's own interpolation differs from the reference's "+0.1" scale-factor form, dinov2_extractor.py:94-118;
    the oracle's resample is pinned to the reference's ``_fix_pos_enc`` by tests/golden/extractor_pieces.npz), so HF does no
    interpolation of its own and witnesses everything else."""
    import transformers
    hf_cfg = transformers.Dinov2Config(hidden_size=cfg.dim, num_hidden_layers=cfg.blocks_run,
                                       num_attention_heads=cfg.heads, mlp_ratio=4, image_size=cfg.img_size,
                                       patch_size=cfg.patch, layer_norm_eps=cfg.ln_eps, hidden_act="gelu",
                                       qkv_bias=True, layerscale_value=1.0, use_swiglu_ffn=False,
                                       hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                                       drop_path_rate=0.0)
    model = transformers.Dinov2Model(hf_cfg).eval()
    own = model.state_dict()
    m = {"embeddings.cls_token": sd["cls_token"], "embeddings.position_embeddings": vit_ref.resample_pos_embed(sd["pos_embed"], cfg.grid),
         "embeddings.patch_embeddings.projection.weight": sd["patch_embed.proj.weight"],
         "embeddings.patch_embeddings.projection.bias": sd["patch_embed.proj.bias"],
         "embeddings.mask_token": own["embeddings.mask_token"]}
    d = cfg.dim
    for i in range(cfg.blocks_run):
        s, t = f"blocks.{i}.", f"encoder.layer.{i}."
        qw, qb = sd[s + "attn.qkv.weight"], sd[s + "attn.qkv.bias"]
        for j, nm in enumerate(("query", "key", "value")):
            m[t + f"attention.attention.{nm}.weight"] = qw[j * d:(j + 1) * d]
            m[t + f"attention.attention.{nm}.bias"] = qb[j * d:(j + 1) * d]
        m[t + "attention.output.dense.weight"] = sd[s + "attn.proj.weight"]
        m[t + "attention.output.dense.bias"] = sd[s + "attn.proj.bias"]
        m[t + "norm1.weight"] = sd[s + "norm1.weight"]
        m[t + "norm1.bias"] = sd[s + "norm1.bias"]
        m[t + "norm2.weight"] = sd[s + "norm2.weight"]
        m[t + "norm2.bias"] = sd[s + "norm2.bias"]
        m[t + "layer_scale1.lambda1"] = sd[s + "ls1.gamma"]
        m[t + "layer_scale2.lambda1"] = sd[s + "ls2.gamma"]
        m[t + "mlp.fc1.weight"] = sd[s + "mlp.fc1.weight"]
        m[t + "mlp.fc1.bias"] = sd[s + "mlp.fc1.bias"]
        m[t + "mlp.fc2.weight"] = sd[s + "mlp.fc2.weight"]
        m[t + "mlp.fc2.bias"] = sd[s + "mlp.fc2.bias"]
    missing, unexpected = model.load_state_dict(m, strict=False)
    assert not [k for k in missing if not k.startswith("layernorm.")], missing
    assert not unexpected
    x = vit_ref.preprocess_u8(frames, cfg.mean, cfg.std)
    with torch.no_grad():
        return model(pixel_values=x, output_hidden_states=True).hidden_states


def test_forward_matches_hf_dinov2_layerscale():
    pytest.importorskip("transformers")
    cfg = _tiny_cfg(True)
    sd = weights.synthetic_state_dict(cfg, 2)
    frames = _frames(cfg)
    hs = _hf_dinov2_hidden_states(cfg, sd, frames)
    mine = _oracle_tokens(cfg, sd, frames, return_all=True)
    for a, b in zip(hs, mine):
        assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max()))


def test_forward_matches_hf_dinov2_at_vits14_308_width():
    """The reference's shipped configuration (config.yaml: DINOv2 ViT-S/14 at 308²): 384 wide, 6 heads, LayerScale gains != 1,
    the stored 37 x 37 position grid resampled to 22 x 22 (485 tokens); three blocks at full width."""
    pytest.importorskip("transformers")
    cfg = dataclasses.replace(config.baseline_config("vits14_308"), depth=3, layer=2)
    assert cfg.layerscale and cfg.native_grid == 37 and cfg.grid == 22
    sd = weights.synthetic_state_dict(cfg, 6)
    assert float((sd["blocks.0.ls1.gamma"] - 1).abs().max()) > 0.1           # gains that would show a misplaced LayerScale
    frames = _frames(cfg, n=2, seed=5)
    hs = _hf_dinov2_hidden_states(cfg, sd, frames)
    mine = _oracle_tokens(cfg, sd, frames, return_all=True)
    assert len(hs) == len(mine) == 4
    for a, b in zip(hs, mine):
        assert a.shape == b.shape == (2, 485, 384)
        assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max()))


def test_descriptor_drops_cls_and_bins():
    cfg = _tiny_cfg(False)
    sd = weights.synthetic_state_dict(cfg, 3)
    frames = _frames(cfg, 1)
    toks = _oracle_tokens(cfg, sd, frames)
    kw = dict(patch=cfg.patch, stride=cfg.stride, heads=cfg.heads, layer=cfg.layer, mean=cfg.mean, std=cfg.std)
    d = vit_ref.extract_descriptors(sd, frames, **kw)
    assert d.shape == (1, 1, cfg.tokens, cfg.dim)
    assert torch.equal(d[0, 0], toks[0, 1:])
    b = vit_ref.extract_descriptors(sd, frames, bin=True, **kw)
    assert b.shape == (1, 1, cfg.tokens, 9 * cfg.dim)
    centre = b[0, 0, :, 4 * cfg.dim:5 * cfg.dim]
    assert torch.equal(centre, toks[0, 1:])


def test_facets_reassemble_the_attention_of_the_hooked_block():
    """extract_facet's layout (index d*H + h, cls dropped — dinov2_extractor.py:326-334) is pinned by putting q, k, v back
    together: softmax(q k^T / 8) v through the block's proj must reproduce the attention branch of blocks[layer] for the
    patch-token queries restricted to patch-token keys... so instead the facets are compared with a direct reshape of
    qkv(norm1(x)) written the reference's way (reshape(B,N,3,H,hd).permute(2,0,3,1,4))."""
    import dataclasses
    import torch.nn.functional as F
    from vitvs_amd import config, synth, weights
    cfg = dataclasses.replace(config.vit_config("dino_vits16", 64), dim=128, depth=2, heads=2, layer=1)
    cfg = dataclasses.replace(cfg, native_grid=cfg.grid)
    sd = weights.synthetic_state_dict(cfg, 3, affine_jitter=True)
    frames = np.stack(synth.frame_pair(cfg.img_size, 9))
    kw = dict(patch=cfg.patch, stride=cfg.stride, heads=cfg.heads, layer=cfg.layer, mean=cfg.mean, std=cfg.std)
    x_in = vit_ref.block_tokens(sd, frames, return_all=True, **kw)[cfg.layer]
    p = f"blocks.{cfg.layer}."
    y = F.layer_norm(x_in, (cfg.dim,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
    B, N, C = y.shape
    qkv = F.linear(y, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(B, N, 3, cfg.heads, C // cfg.heads).permute(2, 0, 3, 1, 4)
    for idx, facet in enumerate(("query", "key", "value")):
        x = qkv[idx][:, :, 1:, :]                                                    # the reference's hook + cls drop
        want = x.permute(0, 2, 3, 1).flatten(start_dim=-2, end_dim=-1).unsqueeze(dim=1)   # its flatten, verbatim semantics
        got = vit_ref.extract_facet(sd, frames, facet=facet, **kw)
        assert got.shape == (2, 1, cfg.tokens, cfg.dim)
        assert torch.equal(got, want)
        h, d = 1, 5                                                                   # spot check of the index rule
        assert torch.equal(got[0, 0, :, d * cfg.heads + h], qkv[idx][0, h, 1:, d])
