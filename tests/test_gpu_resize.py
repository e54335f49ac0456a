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
