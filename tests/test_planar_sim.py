"""The closed-loop test's scene simulator (tests/planar_sim.py) and the CPU oracle in closed loop, without a GPU:
the geometry conventions tests/test_gpu_loop.py relies on (pinhole render by homography, uint16 depth, the published twist
integrated as a body twist with the Gazebo remap undone) are checked on their own, and the oracle's own update — CPU forward,
the reference's correspondence, draw and control law — is shown to drive the simulated camera towards the goal."""
import numpy as np
import torch
from PIL import Image

import vitvs_amd  # noqa: F401
from vitvs_amd import config, synth, weights
from oracle import servo_ref as sr
from oracle import vit_ref
from planar_sim import CameraSim, PlanarScene, quat_xyzw, rodrigues


class _Sink:
    def image_callback_rgb(self, x):
        self.rgb = x

    def image_callback_depth(self, z):
        self.z = z


def test_render_geometry_and_twist_integration():
    params = config.ServoParams(dino_input_size=224, use_feature_binning=False)
    tex = synth.texture(128, 11)
    scene = PlanarScene(tex, 1.6 / 128, params, plane_z=0.61)
    rgb, depth = scene.render(np.eye(3), np.zeros(3))
    assert rgb.shape == (480, 640, 3) and rgb.dtype == np.uint8 and depth.shape == (480, 640) and depth.dtype == np.uint16
    assert np.all(depth == 610)                                  # fronto-parallel plane: depth along the optical axis
    # the principal ray hits the texture centre
    c = tex[63:65, 63:65].astype(np.float64).mean(axis=(0, 1))
    assert np.abs(rgb[240, 320].astype(np.float64) - c).max() <= 24.0
    # moving the camera along +x by one camera pixel's footprint shifts the image content by one pixel to the LEFT
    step = 0.61 / params.f_x
    moved, _ = scene.render(np.eye(3), np.array([step * 8, 0.0, 0.0]))
    a, b = rgb[100:380, 108:540].astype(np.float64), moved[100:380, 100:532].astype(np.float64)
    assert np.abs(a - b).mean() <= 1.0
    # approaching the plane by 10 cm reads 510 mm
    _, near = scene.render(np.eye(3), np.array([0.0, 0.0, 0.10]))
    assert np.all(near == 510)
    # a tilt makes the depth vary monotonically across the image
    _, tilted = scene.render(rodrigues(np.array([0.0, np.deg2rad(10.0), 0.0])), np.zeros(3))
    assert tilted[240, 600] != tilted[240, 40] and np.all(np.diff(tilted[240].astype(np.int64)) * np.sign(int(tilted[240, 600]) - int(tilted[240, 40])) >= -1)
    # twist integration: publish_twist's remap (lin = (v2, -v0, -v1), ang = (w2, -w0, -w1)) is undone, body frame
    sim = CameraSim(scene, _Sink(), rodrigues(np.array([0.0, 0.0, np.pi / 2])), np.zeros(3), dt=1.0)
    lin, ang = sr.twist_remap(np.array([0.1, 0.0, 0.0, 0.0, 0.0, 0.0]), 1.0)       # camera moves along ITS x
    sim.apply_twist(lin, ang)
    np.testing.assert_allclose(sim.t, [0.0, 0.1, 0.0], atol=1e-12)                  # its x is the world's y after the 90 degree yaw
    lin, ang = sr.twist_remap(np.array([0.0, 0.0, 0.0, 0.0, 0.0, 0.2]), 1.0)
    sim.apply_twist(lin, ang)
    np.testing.assert_allclose(sim.R, rodrigues(np.array([0.0, 0.0, np.pi / 2 + 0.2])), atol=1e-12)
    q = quat_xyzw(sim.R)
    np.testing.assert_allclose(np.linalg.norm(q), 1.0, atol=1e-12)
    np.testing.assert_allclose(2 * np.arctan2(np.linalg.norm(q[:3]), q[3]), np.pi / 2 + 0.2, atol=1e-12)


def test_cpu_oracle_closes_the_loop_on_the_simulated_scene():
    """60 updates of the oracle (ViT-S/16 on the CPU, the reference's draw with its seed, EMA, twist remap) from the 5 cm /
    5 degree offset of the GPU test: the feature error shrinks and the camera moves towards the goal pose."""
    cfg = config.baseline_config("vits16_224")
    params = config.ServoParams(dino_input_size=cfg.img_size, use_feature_binning=False)
    sd = weights.synthetic_state_dict(cfg, 0)
    scene = PlanarScene(synth.texture(128, 11), 1.6 / 128, params, plane_z=0.61)
    sink = _Sink()
    axis = np.array([0.3, -0.4, 0.85])
    axis /= np.linalg.norm(axis)
    direction = np.array([0.6, -0.5, 0.6])
    direction /= np.linalg.norm(direction)
    sim = CameraSim(scene, sink, rodrigues(axis * np.deg2rad(5.0)), direction * 0.05, dt=0.5)

    def tokens(rgb):
        small = np.array(Image.fromarray(rgb).resize((cfg.img_size, cfg.img_size)))
        return vit_ref.block_tokens(sd, small[None], patch=cfg.patch, stride=cfg.stride, heads=cfg.heads, layer=cfg.layer,
                                    mean=cfg.mean, std=cfg.std)[0, 1:]
    goal = tokens(scene.render(np.eye(3), np.zeros(3))[0])
    ema = sr.Ema(params.ema_alpha)
    gen = torch.Generator().manual_seed(121)
    errs = []
    for _ in range(60):
        sim.sense()
        out = sr.servo_update(goal, tokens(sink.rgb), sink.z, num_pairs=params.num_pairs, input_size=cfg.img_size, u_max=params.u_max,
                              v_max=params.v_max, fx=params.f_x, fy=params.f_y, lam=params.lambda_, generator=gen, exact_order=False)
        assert out["status"] == "ok"
        errs.append(float(np.linalg.norm(out["e"])))
        lin, ang = sr.twist_remap(ema.update(out["v_c"]), params.max_velocity)
        sim.apply_twist(lin, ang)
    pos_cm = float(np.linalg.norm(sim.t) * 100)
    rot_deg = float(np.degrees(np.arccos(np.clip((np.trace(sim.R) - 1) / 2, -1, 1))))
    assert np.mean(errs[-10:]) <= 0.7 * np.mean(errs[:5]) and pos_cm < 4.5 and rot_deg < 4.5, (errs[:3], errs[-3:], pos_cm, rot_deg)
