"""The oracle's post-ViT half against the REFERENCE'S OWN FUNCTIONS on random inputs — in the build container only.

tests/test_oracle_vs_golden.py pins oracle/servo_ref.py on the committed fixtures (generated once by oracle/make_golden.py from the
reference's functions).  Where /root/reference is present (the build container; never the GPU box: skipped there) this file goes
further: the reference's functions are loaded from its source files by AST (oracle/ref_extract.py; no module import, nothing copied)
and run side by side with the oracle on fresh random inputs — every grid from 4 x 4 to 16 x 16, one to all mutual nearest
neighbours (fewer than four matches included), identical frames (the same-image shortcut), depth images with holes, other camera parameters, num_pairs above and below
the number of candidates — with the same torch RNG seed on both sides, so also the DRAW must agree:

  chunk_cosine_sim, find_correspondences_batch     vitvs_v2.py:49-155
  calculate_uv, transform_to_real_world, get_depth, calculate_interaction_matrix, pinv, EMA     :325-343, 525-659
  ViTExtractor._log_bin                            dinov2_extractor.py:265-311
"""
import numpy as np
import pytest
import torch

import vitvs_amd  # noqa: F401
from vitvs_amd import config, synth
from oracle import ref_extract as rx
from oracle import servo_ref as sr
from oracle import vit_ref

pytestmark = pytest.mark.skipif(not rx.available(), reason="the reference tree is not on this machine (GPU box)")


def _descriptors(rng, t, d, kind):
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    d1 = torch.randn(t, d, generator=g)
    if kind == "same":                                   # identical frames: mean(sim_1) > 0.99
        return d1, d1 + 1e-3 * torch.randn(t, d, generator=g)
    if kind == "funnel":                                 # exactly 1 .. 3 mutual NNs: the "< 4 matches" quirk of calculate_uv (:539-541)
        from oracle.make_golden import _funnel_descriptors
        return _funnel_descriptors(t, int(rng.integers(1, 4)), int(rng.integers(1 << 20)))
    # the true match's cosine is 1 / sqrt(1 + sigma^2) against N(0, 1 / d) for the others: sigma in units of sqrt(d)
    sigma = {"few": 1 / 1.2, "some": 1 / 2.5, "many": 1 / 6.0}[kind] * float(np.sqrt(d))
    return d1, d1 + sigma * torch.randn(t, d, generator=g)


@pytest.mark.parametrize("seed", range(50))
def test_post_vit_path_equals_the_reference_on_random_inputs(seed):
    from oracle.make_golden import reference_post_vit
    rng = np.random.default_rng(31000 + seed)
    grid = int(rng.integers(4, 17))
    t, d = grid * grid, int(rng.choice([32, 96, 384]))
    kind = ["few", "some", "many", "same", "funnel"][seed % 5]
    params = config.ServoParams(num_pairs=int(rng.integers(4, 49)), u_max=int(rng.choice([640, 1280])), v_max=int(rng.choice([480, 720])),
                                f_x=float(rng.uniform(300, 900)), f_y=float(rng.uniform(300, 900)), lambda_=float(rng.uniform(0.01, 0.5)),
                                ema_alpha=float(rng.uniform(0.1, 0.9)), dino_input_size=int(rng.choice([224, 308, 448, 518])))
    d1, d2 = _descriptors(rng, t, d, kind)
    depth = rng.integers(200, 3000, size=(params.v_max, params.u_max)).astype(np.uint16)
    depth.reshape(-1)[rng.integers(0, depth.size, size=depth.size // 5)] = 0          # holes -> the 100 m sentinel
    draw_seed = int(rng.integers(1 << 30))
    ref = reference_post_vit(d1, d2, depth, params, params.dino_input_size, seed=draw_seed)
    gen = torch.Generator().manual_seed(draw_seed)
    torch.manual_seed(draw_seed)                                                        # (the oracle draws from `generator`)
    out = sr.servo_update(d1, d2, depth, num_pairs=params.num_pairs, input_size=params.dino_input_size, u_max=params.u_max,
                          v_max=params.v_max, fx=params.f_x, fy=params.f_y, lam=params.lambda_, generator=gen)
    corr = out["corr"]
    assert np.array_equal(corr["nn_1"].numpy(), ref["nn_1"]) and np.array_equal(corr["nn_2"].numpy(), ref["nn_2"])
    assert np.array_equal(corr["sim_1"].numpy(), ref["sim_1"])                          # bit-exact fp32 similarities
    if int(ref["status"]) == 1:
        assert out["status"] == "no_correspondence"                                     # (None, None, None): every token a mutual NN
        assert int((corr["nn_2"][corr["nn_1"]] == torch.arange(t)).sum()) == t
        return
    assert np.array_equal(corr["points1"].numpy(), ref["points1"]) and np.array_equal(corr["points2"].numpy(), ref["points2"])
    assert np.array_equal(corr["sim"].reshape(-1).numpy(), ref["sim_selected"])
    if kind == "same":
        assert bool(corr["same_image"]) and float(ref["mean_sim_1"]) > 0.99              # the shortcut at vitvs_v2.py:84-101
    assert bool(corr["same_image"]) == (float(ref["mean_sim_1"]) > 0.99)
    assert np.array_equal(out["s_uv_star"], ref["s_uv_star"]) and np.array_equal(out["s_uv"], ref["s_uv"])
    assert np.array_equal(out["Z"], ref["Z"])
    np.testing.assert_array_equal(out["L"], ref["L"])                                   # the same fp64 expressions: bit-exact
    np.testing.assert_array_equal(out["e"], ref["e"])
    np.testing.assert_array_equal(out["v_c"], ref["v_c"])
    assert (out["status"] == "ok") == (int(ref["status"]) == 0)
    ema = sr.Ema(params.ema_alpha)
    np.testing.assert_array_equal(ema.update(out["v_c"]), ref["ema_first"])
    np.testing.assert_array_equal(ema.update(0.5 * out["v_c"]), ref["ema_second"])


@pytest.mark.parametrize("grid,d", [(4, 8), (7, 16), (14, 64), (22, 32), (37, 8)])
def test_log_bin_equals_the_reference(grid, d):
    cls = rx.load_log_bin()
    ext = cls()
    g = torch.Generator().manual_seed(grid * 100 + d)
    toks = torch.randn(2, 1, grid * grid, d, generator=g)                                # the extractor's layout: B x h x t x d
    ext.num_patches = (grid, grid)
    ext.device = "cpu"
    want = ext._log_bin(toks)                                                            # hierarchy = 1, the extractor's default
    got = vit_ref.log_bin(toks[:, 0], grid)
    assert want.shape == (2, 1, grid * grid, 9 * d)
    assert torch.equal(got, want[:, 0])
