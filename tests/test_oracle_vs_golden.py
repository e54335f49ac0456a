"""The CPU oracle (oracle/servo_ref.py, oracle/vit_ref.py) against the fixtures produced by
the reference's own functions (oracle/make_golden.py).  Integer outputs bit-exact, float64
outputs to 1e-12, fp32 similarities bit-exact (same torch ops in the same order)."""
import numpy as np
import pytest
import torch

import vitvs_amd  # noqa: F401
from vitvs_amd import config, synth, weights
from oracle import servo_ref as sr
from oracle import vit_ref
from conftest import golden_case, load_golden

CORR_CASES = ["partial", "short", "tiny", "all_mutual", "grid14", "same_image"]


def _run_oracle(case, depth):
    d1 = torch.from_numpy(case["desc1"])
    d2 = torch.from_numpy(case["desc2"])
    p = config.ServoParams()
    torch.manual_seed(121)
    return sr.servo_update(d1, d2, depth, num_pairs=int(case["num_pairs"]), input_size=int(case["input_size"]),
                           u_max=p.u_max, v_max=p.v_max, fx=p.f_x, fy=p.f_y, lam=p.lambda_)


@pytest.mark.parametrize("name", CORR_CASES)
def test_correspondence_and_law_match_reference(name):
    case = golden_case(load_golden("corr_cases.npz"), name)
    out = _run_oracle(case, synth.depth_pattern())
    corr = out["corr"]
    assert np.array_equal(corr["nn_1"].numpy(), case["nn_1"])
    assert np.array_equal(corr["nn_2"].numpy(), case["nn_2"])
    assert np.array_equal(corr["sim_1"].numpy(), case["sim_1"])  # bit-exact fp32
    status = int(case["status"])
    if status == 1:
        assert out["status"] == "no_correspondence"
        return
    assert np.array_equal(corr["points1"].numpy(), case["points1"])
    assert np.array_equal(corr["points2"].numpy(), case["points2"])
    assert np.array_equal(out["s_uv_star"], case["s_uv_star"])
    assert np.array_equal(out["s_uv"], case["s_uv"])
    assert np.array_equal(out["Z"], case["Z"])
    np.testing.assert_allclose(out["L"], case["L"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(out["v_c"], case["v_c"], rtol=1e-12, atol=1e-18)
    assert (out["status"] == "too_few") == (status == 2)


def test_quirk_all_mutual_returns_none():
    case = golden_case(load_golden("corr_cases.npz"), "all_mutual")
    assert int(case["status"]) == 1
    t = case["nn_1"].shape[0]
    assert np.array_equal(case["nn_2"][case["nn_1"]], np.arange(t))


def test_quirk_zero_padding_rows():
    case = golden_case(load_golden("corr_cases.npz"), "short")
    k = case["points1"].shape[0]
    assert 4 <= k < int(case["num_pairs"])
    assert np.all(case["s_uv"][k:] == 0) and np.all(case["s_uv_star"][k:] == 0)
    assert case["L"].shape == (2 * int(case["num_pairs"]), 6)


def test_quirk_below_four_matches_is_all_zero():
    case = golden_case(load_golden("corr_cases.npz"), "tiny")
    assert int(case["status"]) == 2
    assert np.all(case["s_uv"] == 0) and np.all(case["s_uv_star"] == 0)
    assert np.all(case["v_c"] == 0)


def test_ema_matches_reference():
    case = golden_case(load_golden("corr_cases.npz"), "partial")
    ema = sr.Ema(config.ServoParams().ema_alpha)
    np.testing.assert_allclose(ema.update(case["v_c"]), case["ema_first"], rtol=0, atol=0)
    np.testing.assert_allclose(ema.update(0.5 * case["v_c"]), case["ema_second"], rtol=1e-15)


def test_log_bin_matches_reference():
    blob = load_golden("extractor_pieces.npz")
    for grid in (4, 5):
        x = torch.from_numpy(blob[f"log_bin/g{grid}/x"])
        y = vit_ref.log_bin(x[:, 0], grid)
        assert np.array_equal(y.numpy(), blob[f"log_bin/g{grid}/y"][:, 0])


def test_pos_embed_resample_matches_reference():
    blob = load_golden("extractor_pieces.npz")
    tags = sorted({k.rsplit("/", 1)[0] for k in blob.files if k.startswith("pos/")})
    assert tags
    for tag in tags:
        pe = torch.from_numpy(blob[tag + "/pos_embed"])
        grid = int(blob[tag + "/grid"])
        out = vit_ref.resample_pos_embed(pe, grid)
        assert np.array_equal(out.numpy(), blob[tag + "/out"])
        mine = weights.resample_pos_embed(pe, grid)
        assert np.array_equal(mine.numpy(), blob[tag + "/out"][0])


E2E = [("vits16_224", True), ("vitb16_224", True), ("vits14_308", True)]


@pytest.mark.parametrize("key,binned", E2E)
def test_end_to_end_fixture(key, binned):
    """Oracle ViT + oracle post-ViT path reproduce the fixture (whose post-ViT half was
    computed by the reference's functions), and the fixture meets the acceptance rule."""
    blob = load_golden(f"e2e_{key}.npz")
    cfg = config.baseline_config(key)
    sd = weights.synthetic_state_dict(cfg, int(blob["weight_seed"]))
    assert int(blob["frame_seed"]) == synth.ACCEPTED_FRAME_SEEDS[key]
    des, cur = synth.frame_pair(cfg.img_size, int(blob["frame_seed"]))
    assert [int(des.astype(np.int64).sum()), int(cur.astype(np.int64).sum())] == list(blob["frames_checksum"])
    if "I_des" in blob.files:
        assert np.array_equal(des, blob["I_des"]) and np.array_equal(cur, blob["I_cur"])
    toks = vit_ref.block_tokens(sd, np.stack([des, cur]), patch=cfg.patch, stride=cfg.stride, heads=cfg.heads,
                                layer=cfg.layer, mean=cfg.mean, std=cfg.std)
    np.testing.assert_allclose(toks[:, ::37, ::97].numpy(), blob["token_probe"], rtol=1e-4, atol=1e-5)
    p = config.ServoParams(dino_input_size=cfg.img_size)
    for tag in (("plain", "binned") if binned else ("plain",)):
        case = golden_case(blob, tag)
        t = cfg.tokens
        mutual = int((case["nn_2"][case["nn_1"]] == np.arange(t)).sum())
        assert 4 <= mutual < t and float(case["mean_sim_1"]) <= 0.99
        strict = float(case["margin_rows"]) >= 1e-4 and float(case["margin_cols"]) >= 1e-4
        assert bool(case["strict"]) == strict
        if (key, tag) in (("vits16_224", "plain"), ("vitb16_224", "plain"), ("vits16_224", "binned"),
                          ("vitb16_224", "binned"), ("vits14_308", "binned")):
            assert strict, "headline fixtures must meet the acceptance rule (SURVEY §8(d))"
        d = toks[:, 1:]
        if tag == "binned":
            d = vit_ref.log_bin(d, cfg.grid)
        torch.manual_seed(121)
        out = sr.servo_update(d[0], d[1], synth.depth_pattern(), num_pairs=p.num_pairs, input_size=cfg.img_size,
                              u_max=p.u_max, v_max=p.v_max, fx=p.f_x, fy=p.f_y, lam=p.lambda_)
        assert np.array_equal(out["corr"]["nn_1"].numpy(), case["nn_1"])
        assert np.array_equal(out["corr"]["nn_2"].numpy(), case["nn_2"])
        assert np.array_equal(out["s_uv"], case["s_uv"]) and np.array_equal(out["s_uv_star"], case["s_uv_star"])
        np.testing.assert_allclose(out["v_c"], case["v_c"], rtol=1e-10)


def test_rig8_fixture_pairs_meet_the_acceptance_rule_and_match_the_oracle():
    """BASELINE.json configs[3] (8-camera rig): 8 accepted ViT-B/16 pairs, pair 0 = the headline fixture's pair."""
    blob = load_golden("rig8_vitb16_224.npz")
    seeds = [int(s) for s in blob["frame_seeds"]]
    assert len(seeds) == 8 and seeds == list(synth.RIG8_FRAME_SEEDS) and seeds[0] == synth.ACCEPTED_FRAME_SEEDS["vitb16_224"]
    head = golden_case(load_golden("e2e_vitb16_224.npz"), "plain")
    pair0 = golden_case(blob, "pair0")
    assert np.array_equal(pair0["nn_1"], head["nn_1"]) and np.array_equal(pair0["v_c"], head["v_c"])
    cfg = config.baseline_config("vitb16_224")
    sd = weights.synthetic_state_dict(cfg, int(blob["weight_seed"]))
    p = config.ServoParams(dino_input_size=cfg.img_size)
    for i in (1, 7):                                   # two of the pairs through the whole oracle (CPU time)
        case = golden_case(blob, f"pair{i}")
        assert min(float(case["margin_rows"]), float(case["margin_cols"])) >= 1e-4
        des, cur = synth.frame_pair(cfg.img_size, seeds[i])
        toks = vit_ref.block_tokens(sd, np.stack([des, cur]), patch=cfg.patch, stride=cfg.stride, heads=cfg.heads,
                                    layer=cfg.layer, mean=cfg.mean, std=cfg.std)[:, 1:]
        torch.manual_seed(121)
        out = sr.servo_update(toks[0], toks[1], synth.depth_pattern(), num_pairs=p.num_pairs, input_size=cfg.img_size,
                              u_max=p.u_max, v_max=p.v_max, fx=p.f_x, fy=p.f_y, lam=p.lambda_)
        assert np.array_equal(out["corr"]["nn_1"].numpy(), case["nn_1"])
        assert np.array_equal(out["corr"]["points1"].numpy(), case["points1"])
        np.testing.assert_allclose(out["v_c"], case["v_c"], rtol=1e-10)


def test_rotation_fixture_matches_the_oracle_draws_and_scores():
    """find_and_set_best_pose (vitvs_v2.py:1151-1189): four views, num_pairs = 48, consecutive draws, mean similarity."""
    blob = load_golden("rotation_vits16_224.npz")
    cfg = config.baseline_config("vits16_224")
    sd = weights.synthetic_state_dict(cfg, int(blob["weight_seed"]))
    des, cur = synth.frame_pair(cfg.img_size, int(blob["frame_seed"]))
    views = [np.rot90(cur, int(k)).copy() for k in blob["rot90_k"]]
    toks = vit_ref.block_tokens(sd, np.stack([des] + views), patch=cfg.patch, stride=cfg.stride, heads=cfg.heads,
                                layer=cfg.layer, mean=cfg.mean, std=cfg.std)[:, 1:]
    torch.manual_seed(121)
    scores = []
    for i in range(4):
        corr = sr.find_correspondences(toks[0], toks[1 + i], num_pairs=int(blob["num_pairs"]))
        assert np.array_equal(corr["points1"].numpy(), blob[f"view{i}/points1"])
        assert np.array_equal(corr["points2"].numpy(), blob[f"view{i}/points2"])
        scores.append(corr["sim"].mean().item())
        assert scores[-1] == float(blob[f"view{i}/score"])
    assert int(np.argmax(scores)) == int(blob["best"]) == 1          # the un-rotated view
