"""Host-side mirror of the reference interface (vit-vs_amd/servo.py) on CPU: the selection glue must reproduce
the reference's draw (golden fixtures) when the GPU stage is replaced by the oracle's arg-max tables."""
import numpy as np
import pytest
import torch

import vitvs_amd  # noqa: F401
from vitvs_amd import servo
from oracle import servo_ref as sr
from conftest import golden_case, load_golden


class _OracleBackedEngine:
    """Stands in for Engine.correspond only (similarities + arg-max from the CPU oracle)."""

    def correspond(self, d1, d2):
        sim = sr.cosine_matrix(d1, d2)
        sim_1, nn_1, _, nn_2 = sr.nearest_neighbours(sim)
        return nn_1.int(), nn_2.int(), sim_1


@pytest.mark.parametrize("name", ["partial", "short", "tiny", "grid14", "all_mutual", "same_image"])
def test_find_correspondences_batch_reproduces_reference_draw(name):
    case = golden_case(load_golden("corr_cases.npz"), name)
    d1 = torch.from_numpy(case["desc1"])[None, None]
    d2 = torch.from_numpy(case["desc2"])[None, None]
    torch.manual_seed(121)
    p1, p2, sim = servo.find_correspondences_batch(_OracleBackedEngine(), d1, d2, num_pairs=int(case["num_pairs"]))
    if int(case["status"]) == 1:
        assert p1 is None and p2 is None and sim is None
        return
    assert np.array_equal(p1.numpy(), case["points1"]) and np.array_equal(p2.numpy(), case["points2"])
    np.testing.assert_array_equal(sim.reshape(-1).numpy(), case["sim_selected"])


def test_candidate_order_is_mutual_nn_set():
    case = golden_case(load_golden("corr_cases.npz"), "grid14")
    nn1, nn2 = torch.from_numpy(case["nn_1"]).long(), torch.from_numpy(case["nn_2"]).long()
    cand = servo._candidate_order(nn1, nn2, 14)
    mutual = torch.nonzero(nn2[nn1] == torch.arange(196)).flatten()
    assert sorted(cand.tolist()) == mutual.tolist()


def test_ema_and_twist_match_reference_fixture():
    case = golden_case(load_golden("corr_cases.npz"), "partial")
    state = [None] * 6
    np.testing.assert_array_equal(servo.ema_update(state, case["v_c"], 0.8), case["ema_first"])
    np.testing.assert_allclose(servo.ema_update(state, 0.5 * case["v_c"], 0.8), case["ema_second"], rtol=1e-15)
    lin, ang = servo.twist_from_velocity([0.1, -0.2, 3.0, 0.4, -0.5, 0.6], 1.0)
    assert lin == (1.0, -0.1, 0.2) and ang == (0.6, -0.4, 0.5)
    assert (lin, ang) == sr.twist_remap([0.1, -0.2, 3.0, 0.4, -0.5, 0.6], 1.0)


# ------------------------------------------------------------------------------------------------------------------
# The reference's config.yaml schema (Controller.load_parameters, vitvs_v2.py:272-323)

_CONFIG_KEYS = dict(u_max=1280, v_max=720, lambda_=0.01, min_error=100, max_error=70000, f_x=695.9951, f_y=695.9951,
                    num_pairs=18, image_path="goal.jpg", dino_input_size=518, thresh_filter_keypoints=1,
                    use_feature_binning=False, num_samples=500, num_circles=4, circle_radius_aug=0.08,
                    velocity_convergence_threshold=8e-5, velocity_threshold_translation=5e-19,
                    velocity_threshold_rotation=5e-19, error_threshold_ratio=0.001,
                    error_threshold_absolute_translation=0.1, error_threshold_absolute_rotation=0.1, min_iterations=300,
                    max_iterations=700)


def test_reference_config_schema_is_read_like_load_parameters(tmp_path):
    import yaml
    from vitvs_amd import config
    path = tmp_path / "config.yaml"
    path.write_text(yaml.safe_dump(_CONFIG_KEYS))
    rc = config.load_reference_config(str(path))
    p = rc.servo
    assert (p.u_max, p.v_max, p.num_pairs, p.dino_input_size, p.use_feature_binning) == (1280, 720, 18, 518, False)
    assert p.intrinsics() == (695.9951, 695.9951, 640.0, 360.0)          # principal point = half the image (:282-283)
    assert p.lambda_ == 0.01 and rc.max_iterations == 700 and rc.image_path == "goal.jpg"
    # the optional keys and their defaults in the reference (:287, 316, 319) — ema_alpha falls back to 0.1, not 0.8
    assert p.max_velocity == 1.0 and p.ema_alpha == 0.1 and rc.max_velocity_vector_history == 200
    assert rc.extras["background_thresh"] == 0.5 and rc.extras["num_samples"] == 500 and "u_max" not in rc.extras
    full = dict(_CONFIG_KEYS, ema_alpha=0.8, max_velocity=0.5, max_velocity_vector_history=50, background_thresh=0.005)
    rc = config.load_reference_config(full)
    assert rc.servo.ema_alpha == 0.8 and rc.servo.max_velocity == 0.5 and rc.max_velocity_vector_history == 50
    for missing in ("f_x", "max_iterations", "num_circles", "image_path"):     # indexed with config[...]: KeyError there too
        with pytest.raises(KeyError, match=missing):
            config.load_reference_config({k: v for k, v in _CONFIG_KEYS.items() if k != missing})

    class Stub:
        max_velocity_vector_history = 200
        max_iterations = 1500
    ctl, loop = Stub(), Stub()
    rc.apply(controller=ctl, loop=loop)
    assert ctl.max_velocity_vector_history == 50 and loop.max_iterations == 700


def test_shipped_reference_config_gives_the_package_defaults():
    """In the build container only: the reference's own config.yaml parses to ServoParams' defaults (which cite it)."""
    import os
    from vitvs_amd import config
    shipped = "/root/reference/catkin_ws/ibvs/config/config.yaml"
    if not os.path.exists(shipped):
        pytest.skip("the reference tree is not on this machine")
    rc = config.load_reference_config(shipped)
    assert rc.servo == config.ServoParams()
    assert rc.max_iterations == 1500 and rc.max_velocity_vector_history == 200


# ------------------------------------------------------------------------------------------------------------------
# servo.MultiController's bookkeeping on CPU (a stand-in engine; the device path is tests/test_gpu_multi.py)
class _ScriptedEngine:
    """Batched stand-in for Engine: the twist of camera frame f is a function of the frame's first pixel, status from its second."""
    max_pairs, tokens, max_rows = 8, 16, 48
    device = torch.device("cpu")

    class cfg:
        img_size = 8

    def __init__(self, params):
        self.params = params
        self.calls = []

    def set_frame_size(self, *a):
        return self

    def compute_velocity(self, cur, des, z, K, mode=None, selection=None, des_shared=False, num_pairs=None):
        n = cur.shape[0]
        self.calls.append(dict(n=n, mode=mode, selection=selection, num_pairs=num_pairs, z=np.asarray(z).copy()))
        v = torch.stack([torch.full((6,), float(cur[j, 0, 0, 0]) + 0.25 * float(des[j, 0, 0, 0]), dtype=torch.float64) for j in range(n)])
        st = torch.tensor([int(cur[j, 0, 0, 1]) for j in range(n)], dtype=torch.int32)
        return v, st


def test_multi_controller_keeps_every_cameras_state_apart():
    from vitvs_amd import _lib, config
    params = config.ServoParams(dino_input_size=8, use_feature_binning=False)
    eng = _ScriptedEngine(params)
    frame = lambda a, status=0: np.full((8, 8, 3), 0, np.uint8) + np.array([a, status, 0], np.uint8)   # noqa: E731
    goals = [frame(4 * c) for c in range(3)]
    mc = servo.MultiController(eng, goals, selection="order", generator=torch.Generator().manual_seed(2))
    depth = np.ones((params.v_max, params.u_max), np.uint16)
    # round 1: camera 0 fine, camera 1 reports no correspondence, camera 2 has no image yet
    mc.image_callback_rgb(0, frame(10)); mc.image_callback_depth(0, depth)
    mc.image_callback_rgb(1, frame(20, _lib.STATUS_NO_CORRESPONDENCE)); mc.image_callback_depth(1, depth)
    mc.ibvs()
    call = eng.calls[-1]
    assert call["n"] == 2 and call["mode"] == _lib.SELECT_ORDER and call["selection"].shape == (2, eng.tokens)
    assert sorted(call["selection"][0].tolist()) == list(range(eng.tokens))          # a permutation per camera
    assert not torch.equal(call["selection"][0], call["selection"][1])
    assert np.array_equal(mc.cameras[0].v_c, np.full(6, 10.0)) and mc.cameras[0].feature_failure_count == 0   # EMA: first sample
    assert mc.cameras[1].v_c is None and mc.cameras[1].feature_failure_count == 1 and mc.cameras[1].last_status == 1
    assert mc.cameras[2].v_c is None and mc.cameras[2].last_status is None
    # round 2: camera 0 again (EMA advances), camera 1 recovers (counter back to 0), camera 2 arrives without depth
    mc.image_callback_rgb(0, frame(30))
    mc.image_callback_rgb(1, frame(40))
    mc.image_callback_rgb(2, frame(50))
    mc.ibvs()
    a = params.ema_alpha
    assert np.allclose(mc.cameras[0].v_c, a * 30.0 + (1 - a) * 10.0, rtol=0, atol=1e-15)
    assert np.array_equal(mc.cameras[1].v_c, np.full(6, 40.0 + 0.25 * 4)) and mc.cameras[1].feature_failure_count == 0
    assert mc.cameras[2].v_c is None and mc.cameras[2].last_status == 0               # "Failed to get depth - skipping": v_c stays unset
    assert eng.calls[-1]["n"] == 3 and not eng.calls[-1]["z"][2].any()                # a dummy depth image went in for camera 2
    assert len(mc.cameras[0].velocity_vector_history) == 2 and len(mc.cameras[1].velocity_vector_history) == 1
    assert mc.v_c[2] is None and mc.publish_twist(0) == servo.twist_from_velocity(mc.cameras[0].v_c, params.max_velocity)
    # the 10th consecutive failure of ONE camera raises the reference's error; the others' counters are untouched
    for _ in range(8):
        mc.image_callback_rgb(1, frame(20, _lib.STATUS_NO_CORRESPONDENCE))
        mc.ibvs()
    assert mc.cameras[1].feature_failure_count == 8
    mc.ibvs()
    # ... AFTER every camera of that round has been absorbed: camera 2 comes after the failing camera 1 in the round's loop, and N
    # independent Controllers would each have taken their own step
    mc.image_callback_depth(2, depth)
    mc.image_callback_rgb(2, frame(60))
    before = (len(mc.cameras[0].velocity_vector_history), len(mc.cameras[2].velocity_vector_history))
    with pytest.raises(RuntimeError, match="Persistent feature detection failure"):
        mc.ibvs()
    assert mc.cameras[0].feature_failure_count == 0 and mc.cameras[1].feature_failure_count == 10
    assert (len(mc.cameras[0].velocity_vector_history), len(mc.cameras[2].velocity_vector_history)) == (before[0] + 1, before[1] + 1)
    assert mc.cameras[2].last_status == 0 and mc.cameras[2].v_c is not None
    # explicit ids per round
    mc2 = servo.MultiController(eng, goals[:2])
    mc2.image_callback_rgb(0, frame(1)); mc2.image_callback_rgb(1, frame(2))
    mc2.ibvs(selection=[[1, 2, 3, 4], [5, 6, 7, 8]])
    assert eng.calls[-1]["mode"] == _lib.SELECT_EXPLICIT and [list(s) for s in eng.calls[-1]["selection"]] == [[1, 2, 3, 4], [5, 6, 7, 8]]
    with pytest.raises(ValueError):
        servo.MultiController(eng, goals, selection="reference")
    with pytest.raises(ValueError):
        servo.MultiController(eng, [frame(0)] * 9)


def test_controller_asks_for_a_refused_frame_geometry_only_once():
    """A camera geometry the fused resize cannot take (``Engine.set_frame_size`` raises; the handle keeps its previous geometry) is
    remembered: every attempt builds Pillow's tables, allocates and synchronises the device, so the fallback (the stand-alone
    resize) must not pay for it on every control step."""
    from vitvs_amd import config
    from vitvs_amd.engine import VitvsError

    class _Refusing:
        def __init__(self):
            self.params = config.ServoParams(dino_input_size=8, use_feature_binning=False)
            self.cfg = type("C", (), dict(img_size=8, grid=2))()
            self.asked, self.resized = [], 0
            self.frame_size = (8, 8)

        def set_frame_size(self, h=None, w=None):
            if h is not None and (h, w) != (8, 8):
                self.asked.append((h, w))
                raise VitvsError("camera frame too large for the fused resize")
            return self

        def resize_frames(self, arr):
            self.resized += 1
            return [np.zeros((8, 8, 3), np.uint8)]

    eng = _Refusing()
    ctl = servo.Controller(eng, goal_image=np.zeros((480, 640, 3), np.uint8), selection="order")
    for _ in range(5):
        frames = ctl._path_frames(np.zeros((480, 640, 3), np.uint8), ctl.goal_image)
        assert all(np.asarray(f).shape == (8, 8, 3) for f in frames)
    assert eng.asked == [(480, 640)] and eng.resized == 10          # asked once, resized every time
    ctl._path_frames(np.zeros((240, 320, 3), np.uint8), np.zeros((240, 320, 3), np.uint8))
    assert eng.asked == [(480, 640), (240, 320)]                     # another geometry is another question


# ------------------------------------------------------------------------------------------------------------------
# The Controller's default selection draws on the host between two device calls: the fast form of that draw (integer part in numpy,
# `torch.randperm(n, generator=g)` instead of swapping the global RNG's state) must BE the reference-shaped one.

def _draw_with_global_rng(nn_1, nn_2, sim_1, grid, k, gen):
    """The draw as the reference runs it (vitvs_v2.py:84-141): global RNG, tensors, `_candidate_order`."""
    state = torch.get_rng_state()
    torch.set_rng_state(gen.get_state())
    try:
        t = nn_1.numel()
        if sim_1.mean().item() > 0.99:
            return torch.randperm(t)[:min(k, t)]
        cand = servo._candidate_order(nn_1, nn_2, grid)
        kk = min(k, cand.numel())
        return None if kk == 0 else cand[torch.randperm(cand.numel())[:kk]]
    finally:
        gen.set_state(torch.get_rng_state())
        torch.set_rng_state(state)


def test_host_draw_is_the_reference_shaped_draw_on_random_and_degenerate_tables():
    rng = np.random.default_rng(0)
    for case in range(400):
        g = int(rng.choice([14, 16, 22, 37]))
        t = g * g
        kind = case % 5
        nn1, nn2 = rng.integers(0, t, t), rng.integers(0, t, t)
        if kind == 0:                                        # some mutual pairs
            idx = rng.permutation(t)[:rng.integers(1, t)]
            nn2[nn1[idx]] = idx
        elif kind == 1:                                      # all mutual
            nn1 = rng.permutation(t)
            nn2 = np.argsort(nn1)
        elif kind == 2:                                      # none mutual: the candidates are the smallest displacement's ties
            nn1 = nn2 = (np.arange(t) + 1) % t
        elif kind == 3:
            nn1 = np.minimum(np.arange(t) + g, t - 1)
            nn2 = np.maximum(np.arange(t) - g + rng.integers(-1, 2, t), 0)
        sim = (rng.random(t) * (1.0 if kind != 4 else 0.001) + (0.0 if kind != 4 else 0.995)).astype(np.float32)   # kind 4: same-image shortcut
        k = int(rng.integers(1, 60))
        ga, gb = torch.Generator().manual_seed(case), torch.Generator().manual_seed(case)
        a = _draw_with_global_rng(torch.from_numpy(nn1).long(), torch.from_numpy(nn2).long(), torch.from_numpy(sim), g, k, ga)
        b = servo._draw_like_the_reference(nn1.astype(np.int32), nn2.astype(np.int32), sim, g, k, gb)
        assert (a is None and b is None) or torch.equal(a, b), (case, kind)
        assert torch.equal(ga.get_state(), gb.get_state()), "the draw consumed a different amount of the RNG stream"
        # tensors in: the same function's other branch
        gc = torch.Generator().manual_seed(case)
        c = servo._draw_like_the_reference(torch.from_numpy(nn1).long(), torch.from_numpy(nn2).long(), torch.from_numpy(sim), g, k, gc)
        assert (a is None and c is None) or torch.equal(a, c)
