"""The device resize (vitvs_resize_frames_dev, csrc/resize.hip) against the CPU restatement of Pillow's resample and,
when Pillow is importable, against Pillow itself.  uint8 work: bit-exact."""
import json
import hashlib
import os

import numpy as np
import pytest
import torch

import vitvs_amd  # noqa: F401
from oracle import resize_ref
from vitvs_amd import config, weights
from vitvs_amd.engine import Engine

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "resize_pil.json")


def _engine(key):
    cfg = config.baseline_config(key)
    return Engine(cfg, config.ServoParams(dino_input_size=cfg.img_size), precision="bf16", max_pairs=1), cfg


@pytest.mark.parametrize("key,h,w", [("vitb16_224", 480, 640), ("vits14_308", 480, 640), ("vitb16_224", 720, 1280),
                                     ("vitb16_224", 224, 224), ("vitb16_224", 100, 37), ("vitl14_518", 480, 640)])
def test_device_resize_matches_oracle_and_pillow(key, h, w):
    eng, cfg = _engine(key)
    rng = np.random.default_rng(h + w)
    frames = rng.integers(0, 256, size=(2, h, w, 3), dtype=np.uint8)
    frames[1, ::3, ::5] = 255      # saturating structure in the second frame
    frames[1, 1::3, ::4] = 0
    got = eng.resize_frames(frames).cpu().numpy()
    assert got.shape == (2, cfg.img_size, cfg.img_size, 3) and got.dtype == np.uint8
    for i in range(2):
        assert np.array_equal(got[i], resize_ref.resize_bicubic_u8(frames[i], cfg.img_size))
    try:
        from PIL import Image
    except ImportError:
        return
    for i in range(2):
        assert np.array_equal(got[i], np.asarray(Image.fromarray(frames[i]).resize((cfg.img_size, cfg.img_size))))


def test_device_resize_matches_committed_pillow_digest():
    """Golden vector made with Pillow in the build container (oracle/make_resize_golden.py): seeded 480x640 frame -> 224."""
    gold = json.load(open(GOLD))
    eng, cfg = _engine("vitb16_224")
    rng = np.random.default_rng(gold["seed"])
    frame = rng.integers(0, 256, size=(gold["h"], gold["w"], 3), dtype=np.uint8)
    got = eng.resize_frames(frame).cpu().numpy()[0]
    assert hashlib.sha256(got.tobytes()).hexdigest() == gold["sha256"]
    assert got[: 2, : 4].tolist() == gold["first_pixels"]


def test_resize_then_update_equals_update_on_pillow_frames():
    """Controller-style use: camera frames resized on the device feed the same update as host-resized ones."""
    from vitvs_amd import servo, synth
    eng, cfg = _engine("vitb16_224")
    eng.load_state_dict(weights.synthetic_state_dict(cfg, 0))
    big = [np.kron(synth.frame_pair(cfg.img_size, 77 + i)[i], np.ones((2, 3, 1), np.uint8))[:480, :640] for i in range(2)]
    small_ref = [resize_ref.resize_bicubic_u8(b, cfg.img_size) for b in big]
    z = synth.depth_pattern()
    gen = lambda: torch.Generator().manual_seed(1)   # noqa: E731  (the same visiting order for both updates)
    v1, s1 = servo.compute_velocity(eng, eng.resize_frames(big[1])[0], eng.resize_frames(big[0])[0], z, generator=gen())
    v2, s2 = servo.compute_velocity(eng, small_ref[1], small_ref[0], z, generator=gen())
    assert s1 == s2 and np.array_equal(np.asarray(v1), np.asarray(v2))


# ------------------------------------------------------------------------------------------------------------------
# The resize INSIDE the path (vitvs_set_frame_size): camera frames in, the patch rows are built from the
# pixels PIL would have produced.  Everything downstream of the patch rows is the same launch sequence, so the tokens of
# a camera frame must equal the tokens of its (oracle-)resized image bit for bit, in every precision.

def _camera_frames(n, h, w, seed):
    rng = np.random.default_rng(seed)
    f = rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)
    f[-1, ::3, ::5] = 255          # saturating structure: the intermediate image clips
    f[-1, 1::3, ::4] = 0
    return f


@pytest.mark.parametrize("model,size,stride,h,w,prec", [
    ("vit_base_patch16_224", 224, None, 480, 640, "fp32"), ("vit_base_patch16_224", 224, None, 480, 640, "bf16"),
    ("dinov2_vits14", 308, None, 480, 640, "fp16"), ("vit_base_patch16_224", 224, None, 720, 1280, "bf16"),
    ("dino_vits8", 224, 4, 480, 640, "fp32"),           # overlapping patches: a pixel is rebuilt by up to 4 tokens
    ("dino_vits16", 224, None, 100, 37, "fp32"),        # enlarging on both axes
    ("dino_vits16", 224, None, 224, 640, "bf16"),       # one axis untouched (Pillow still runs its pass: identity taps)
    ("dino_vits16", 64, 8, 1080, 1920, "fp32"),         # 17 x / 30 x reduction: windows of ~70 / 120 taps
])
def test_fused_resize_tokens_equal_tokens_of_the_resized_frames(model, size, stride, h, w, prec):
    cfg = config.vit_config(model, size, stride=stride)
    eng = Engine(cfg, config.ServoParams(dino_input_size=cfg.img_size), precision=prec, max_pairs=1)
    eng.load_state_dict(weights.synthetic_state_dict(cfg, 3))
    frames = _camera_frames(2, h, w, h * 7 + w)
    small = np.stack([resize_ref.resize_bicubic_u8(f, cfg.img_size) for f in frames])
    want = eng.forward_tokens(small).cpu().numpy()
    eng.set_frame_size(h, w)
    assert eng.frame_size == ((h, w) if (h, w) != (size, size) else (size, size))
    got = eng.forward_tokens(frames).cpu().numpy()
    assert np.isfinite(got).all() and np.array_equal(got, want)
    with pytest.raises(Exception):
        eng.forward_tokens(small)                      # the declared geometry is the one that is accepted
    eng.set_frame_size()
    assert np.array_equal(eng.forward_tokens(small).cpu().numpy(), want)


@pytest.mark.parametrize("graph", ["0", "1"])
def test_fused_resize_update_equals_update_on_pillow_frames(graph, monkeypatch):
    """compute_velocity on camera frames (device and host-buffer entry points, recomputed and cached goal) equals the
    update on the host-resized frames bit for bit; also under graph replay, where the geometry is part of the key."""
    from vitvs_amd import _lib, servo, synth
    monkeypatch.setenv("VITVS_GRAPH", graph)
    eng, cfg = _engine("vitb16_224")
    eng.load_state_dict(weights.synthetic_state_dict(cfg, 0))
    big = [np.kron(synth.frame_pair(cfg.img_size, 77 + i)[i], np.ones((2, 3, 1), np.uint8))[:480, :640] for i in range(2)]
    small = [resize_ref.resize_bicubic_u8(b, cfg.img_size) for b in big]
    z = synth.depth_pattern()
    gen = lambda: torch.Generator().manual_seed(1)   # noqa: E731  (the same visiting order for every update)
    v_ref, s_ref = servo.compute_velocity(eng, small[1], small[0], z, generator=gen())
    cam = big[0].shape[:2]                              # 448 x 640
    eng.set_frame_size(*cam)
    for _ in range(2):                                  # the second call replays the captured update when graph == "1"
        v, s = servo.compute_velocity(eng, big[1], big[0], z, generator=gen())
        assert s == s_ref and np.array_equal(np.asarray(v), np.asarray(v_ref))
    eng.set_goal(big[0])                                # cached goal, built from the camera frame
    v, s = servo.compute_velocity(eng, big[1], None, z, generator=gen())
    eng.set_frame_size()
    eng.set_goal(small[0])
    v2, s2 = servo.compute_velocity(eng, small[1], None, z, generator=gen())
    assert s == s2 and np.array_equal(np.asarray(v), np.asarray(v2))
    # host-buffer entry point (vitvs_compute_velocity): frames staged at camera size
    import ctypes as C
    order = torch.randperm(eng.tokens, generator=gen()).to(torch.int32).numpy()
    K = np.asarray(eng.params.intrinsics(), np.float64)

    def host_call(cur, des):
        v_c, st = np.zeros(6, np.float64), np.zeros(1, np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
        cur, des, zz = np.ascontiguousarray(cur), np.ascontiguousarray(des), np.ascontiguousarray(z)
        rc = eng.lib.vitvs_compute_velocity(eng.handle, 1, p(cur), p(des), 0, p(zz), p(K), _lib.SELECT_ORDER, p(order), None,
                                            eng.params.num_pairs, p(v_c), p(st))
        assert rc == 0, _lib.last_error(eng.handle)
        return v_c, int(st[0])
    v_small = host_call(small[1], small[0])
    eng.set_frame_size(*cam)
    v_big = host_call(big[1], big[0])
    assert v_big[1] == v_small[1] and np.array_equal(v_big[0], v_small[0])
    assert eng.lib.vitvs_set_frame_size(eng.handle, 0, 5) != 0           # half a geometry is refused


def test_controller_feeds_camera_frames_straight_into_the_path():
    """Controller.detect_features / ibvs with 448 x 640 camera frames: the engine is switched to the camera geometry (no
    separate resize launch) and the update equals the one on host-resized frames."""
    from vitvs_amd import servo, synth
    eng, cfg = _engine("vitb16_224")
    eng.load_state_dict(weights.synthetic_state_dict(cfg, 0))
    big = [np.kron(synth.frame_pair(cfg.img_size, 5 + i)[i], np.ones((2, 3, 1), np.uint8))[:480, :640] for i in range(2)]
    small = [resize_ref.resize_bicubic_u8(b, cfg.img_size) for b in big]
    z = synth.depth_pattern()
    out = []
    for goal, cur in ((big[0], big[1]), (small[0], small[1])):
        ctl = servo.Controller(eng, goal_image=goal)
        ctl.image_callback_rgb(cur)
        ctl.image_callback_depth(z)
        torch.manual_seed(11)
        out.append(ctl.detect_features())
    assert eng.frame_size == (cfg.img_size, cfg.img_size)          # the second controller saw S x S frames
    (pts_a, sim_a), (pts_b, sim_b) = out
    assert pts_a is not None and np.array_equal(pts_a[0], pts_b[0]) and np.array_equal(pts_a[1], pts_b[1])
    assert torch.equal(sim_a, sim_b)
