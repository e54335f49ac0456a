"""``servo.MultiController``: N cameras — each with its own goal image, EMA state, failure counter and history, i.e. N reference
``Controller``s' worth of state (vitvs_v2.py:224, 325-343, 500-505, 588-632) — served by ONE GPU, either by one batched call per
round (an ``Engine`` with ``max_pairs >= N``) or by one pipelined update per camera (an ``UpdatePipeline``).  BASELINE.json
configs[3] (the 8-camera rig) on one GPU, reachable from the reference's API surface.

Every camera's raw and smoothed ``v_c`` must be those of an independent ``Controller(Engine)`` fed the same frames and the same
draws, bit for bit, round after round; on the rig fixture's pairs with the reference's own draw they are the reference's."""
import numpy as np
import pytest
import torch

import vitvs_amd  # noqa: F401
from vitvs_amd import _lib, config, servo, synth, weights
from vitvs_amd.engine import Engine
from vitvs_amd.pipeline import UpdatePipeline
from conftest import golden_case, load_golden

pytestmark = pytest.mark.gpu

N_CAM, N_ROUNDS = 8, 20


def _camera_streams(cfg, n_cam, n_rounds):
    """Per camera: a goal frame and `n_rounds` current frames (the rig's accepted pairs, the current frame drifting from round to
    round by a few pixels: open loop, the frames do not depend on the twists), one depth image per camera."""
    goals, frames = [], []
    for c in range(n_cam):
        des, cur = synth.frame_pair(cfg.img_size, synth.RIG8_FRAME_SEEDS[c])
        goals.append(des)
        frames.append([np.roll(cur, shift=(r % 5) - 2, axis=1).copy() for r in range(n_rounds)])
    depth = [np.roll(synth.depth_pattern(), 7 * c, axis=1).copy() for c in range(n_cam)]
    return goals, frames, depth


def _run_independent(cfg, params, sd, precision, goals, frames, depth, seed, in_flight=1):
    """N independent Controller(Engine): ibvs() called in camera order every round, drawing from torch's global RNG."""
    engines = [Engine(cfg, params, precision=precision, max_pairs=1).load_state_dict(sd) for _ in goals]
    for e in engines:
        e.set_option("in_flight", in_flight)
    ctls = [servo.Controller(e, goal_image=g, selection="order") for e, g in zip(engines, goals)]
    torch.manual_seed(seed)
    raw, smooth, status = [], [], []
    for r in range(len(frames[0])):
        row_raw, row_s, row_st = [], [], []
        for c, ctl in enumerate(ctls):
            ctl.image_callback_rgb(frames[c][r])
            ctl.image_callback_depth(depth[c])
            ctl.ibvs()
            row_raw.append(np.array(ctl._raw_v, np.float64).copy())
            row_s.append(None if ctl.v_c is None else np.array(ctl.v_c).copy())
            row_st.append(ctl.last_status)
        raw.append(row_raw); smooth.append(row_s); status.append(row_st)
    state = [(list(c.ema_velocities), c.feature_failure_count, [np.array(v) for v in c.velocity_vector_history]) for c in ctls]
    for e in engines:
        e.close()
    return raw, smooth, status, state


def _run_multi(backend, goals, frames, depth, seed, want_features=False):
    mc = servo.MultiController(backend, goals, selection="order")
    torch.manual_seed(seed)
    raw, smooth, status, feats = [], [], [], []
    for r in range(len(frames[0])):
        for c in range(len(goals)):
            mc.image_callback_rgb(c, frames[c][r])
            mc.image_callback_depth(c, depth[c])
        feats.append(mc.ibvs(want_features=want_features))
        raw.append([np.array(c._raw_v, np.float64).copy() for c in mc.cameras])
        smooth.append([None if c.v_c is None else np.array(c.v_c).copy() for c in mc.cameras])
        status.append([c.last_status for c in mc.cameras])
    state = [(list(c.ema_velocities), c.feature_failure_count, [np.array(v) for v in c.velocity_vector_history]) for c in mc.cameras]
    return raw, smooth, status, state, feats


def _same(a, b):
    ra, sa, sta, state_a = a[:4]
    rb, sb, stb, state_b = b[:4]
    assert sta == stb
    for r, (row_a, row_b) in enumerate(zip(ra, rb)):
        for c, (x, y) in enumerate(zip(row_a, row_b)):
            assert np.array_equal(x, y), f"raw v_c differs: round {r}, camera {c}"
    for r, (row_a, row_b) in enumerate(zip(sa, sb)):
        for c, (x, y) in enumerate(zip(row_a, row_b)):
            assert (x is None) == (y is None) and (x is None or np.array_equal(x, y)), f"smoothed v_c differs: round {r}, camera {c}"
    for c, ((ema_a, fail_a, hist_a), (ema_b, fail_b, hist_b)) in enumerate(zip(state_a, state_b)):
        assert ema_a == ema_b and fail_a == fail_b and len(hist_a) == len(hist_b)
        assert all(np.array_equal(x, y) for x, y in zip(hist_a, hist_b))


@pytest.fixture(scope="module")
def rig():
    cfg = config.baseline_config("vitb16_224")
    params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=False)
    sd = weights.synthetic_state_dict(cfg, 0)
    return cfg, params, sd, _camera_streams(cfg, N_CAM, N_ROUNDS)


def test_eight_cameras_in_one_batched_call_equal_eight_controllers(rig):
    """fp32 (the parity mode: the batched many-row tiles and the one-pair tiles sum K in different orders, the arg-max tables are
    the same on these accepted pairs, hence the same draws and bit-identical fp64 twists)."""
    cfg, params, sd, (goals, frames, depth) = rig
    want = _run_independent(cfg, params, sd, "fp32", goals, frames, depth, seed=5)
    eng = Engine(cfg, params, precision="fp32", max_pairs=N_CAM).load_state_dict(sd)
    got = _run_multi(eng, goals, frames, depth, seed=5, want_features=True)
    _same(got, want)
    assert all(s == 0 for row in got[2] for s in row)
    feats = got[4][-1]
    assert len(feats) == N_CAM and all(f[0] is not None and f[0][0].shape == (params.num_pairs, 2) for f in feats)
    eng.close()


@pytest.mark.parametrize("precision", ["bf16", "fp32", "f16x2"])
def test_eight_cameras_through_one_pipeline_equal_eight_controllers(rig, precision):
    """Three updates in flight (the arrangement bench.py measures `value` with): camera i's update is the one-stream update of a
    handle with the same tile plan, so also in the headline dtype every twist is bit-identical."""
    cfg, params, sd, (goals, frames, depth) = rig
    want = _run_independent(cfg, params, sd, precision, goals, frames, depth, seed=9, in_flight=3)
    pipe = UpdatePipeline(cfg, params, sd, precision=precision, depth=3)
    got = _run_multi(pipe, goals, frames, depth, seed=9)
    _same(got, want)
    # a second MultiController on the same pipeline (graphs already captured for other buffers) with features wanted
    got2 = _run_multi(pipe, goals, [f[:3] for f in frames], depth, seed=9, want_features=True)
    _same(got2, tuple(x[:3] if i < 3 else None for i, x in enumerate(want))[:3] + (got2[3],))
    pipe.close()


@pytest.mark.parametrize("backend", ["batched", "pipeline"])
def test_rig_fixture_through_the_multi_camera_adapter(backend):
    """The 8 accepted pairs of tests/golden/rig8_vitb16_224.npz with the reference's own draw (explicit token ids): every camera's
    raw twist is the reference's (<= 1e-9, bar 1e-4), the first smoothed twist equals it (EMA's first-sample rule)."""
    blob = load_golden("rig8_vitb16_224.npz")
    cfg = config.baseline_config("vitb16_224")
    sd = weights.synthetic_state_dict(cfg, int(blob["weight_seed"]))
    params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=False)
    pairs = [synth.frame_pair(cfg.img_size, int(s)) for s in blob["frame_seeds"]]
    sels = [(blob[f"pair{i}/points1"][:, 0] * cfg.grid + blob[f"pair{i}/points1"][:, 1]).astype(np.int32) for i in range(8)]
    be = (Engine(cfg, params, precision="fp32", max_pairs=8).load_state_dict(sd) if backend == "batched"
          else UpdatePipeline(cfg, params, sd, precision="fp32", depth=3))
    mc = servo.MultiController(be, [p[0] for p in pairs])
    for i, p in enumerate(pairs):
        mc.image_callback_rgb(i, p[1])
        mc.image_callback_depth(i, synth.depth_pattern())
    feats = mc.ibvs(selection=sels, want_features=True)
    for i in range(8):
        case = golden_case(blob, f"pair{i}")
        cam = mc.cameras[i]
        assert cam.last_status == 0
        rel = np.linalg.norm(cam._raw_v - case["v_c"]) / np.linalg.norm(case["v_c"])
        assert rel <= 1e-9
        assert np.array_equal(cam.v_c, np.asarray(cam._raw_v, np.float64))
        (s_star, s_), _ = feats[i]
        assert np.array_equal(s_star, case["s_uv_star"]) and np.array_equal(s_, case["s_uv"])
    be.close()


def test_failures_and_missing_inputs_stay_per_camera(rig):
    """A camera without an image is skipped (its state untouched), a camera without depth keeps its stale v_c like the
    reference's ibvs() (vitvs_v2.py:598-619), identical frames (all tokens mutual -> (None, None, None)) count towards THAT
    camera's failure counter only."""
    cfg, params, sd, (goals, frames, depth) = rig
    eng = Engine(cfg, params, precision="fp32", max_pairs=4).load_state_dict(sd)
    mc = servo.MultiController(eng, goals[:4])
    mc.image_callback_rgb(0, frames[0][0]); mc.image_callback_depth(0, depth[0])
    mc.image_callback_rgb(1, goals[1]);     mc.image_callback_depth(1, depth[1])      # same image as its goal
    mc.image_callback_rgb(2, frames[2][0])                                             # no depth
    torch.manual_seed(1)                                                               # camera 3: nothing at all
    out = mc.ibvs(want_features=True)
    assert out[3] is None and mc.cameras[3].last_status is None and mc.cameras[3].v_c is None
    assert mc.cameras[0].last_status == 0 and mc.cameras[0].v_c is not None and mc.cameras[0].feature_failure_count == 0
    # no depth: features are detected on a dummy depth image (detect_features needs none), the law step is skipped: v_c stays unset
    assert mc.cameras[2].last_status == 0 and mc.cameras[2].v_c is None and out[2][0] is not None
    # identical frames take the same-image shortcut of find_correspondences_batch (mean similarity > 0.99): features with zero error
    assert mc.cameras[1].last_status == 0 and np.all(np.asarray(mc.cameras[1]._raw_v) == 0.0)
    eng.close()


@pytest.mark.parametrize("backend", ["batched", "pipeline"])
def test_camera_resolution_frames_through_the_multi_camera_adapter(rig, backend):
    """Cameras deliver 640 x 480 frames (vitvs_v2.py:455-458); the adapter hands them to the engines as they are and the reference's
    PIL resize (:474-475) happens inside the patch-row build (Engine.set_frame_size on every engine of the backend).  Three cameras,
    four rounds: bit-identical to three independent Controller(Engine), which take the same fused path."""
    from PIL import Image
    cfg, params, sd, (goals, frames, depth) = rig
    big = lambda a: np.array(Image.fromarray(a).resize((640, 480), Image.BILINEAR), dtype=np.uint8)   # noqa: E731  (any 640 x 480 content)
    goals3 = [big(g) for g in goals[:3]]
    frames3 = [[big(f) for f in frames[c][:4]] for c in range(3)]
    prec = "fp32"
    want = _run_independent(cfg, params, sd, prec, goals3, frames3, depth[:3], seed=21, in_flight=3 if backend == "pipeline" else 1)
    be = (Engine(cfg, params, precision=prec, max_pairs=3).load_state_dict(sd) if backend == "batched"
          else UpdatePipeline(cfg, params, sd, precision=prec, depth=3))
    got = _run_multi(be, goals3, frames3, depth[:3], seed=21)
    _same(got, want)
    engines = be.engines if backend == "pipeline" else [be]
    assert all(e.frame_size == (480, 640) for e in engines)
    be.close()
