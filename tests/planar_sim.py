"""Test infrastructure: a textured plane seen by a pinhole camera that integrates the published twist.

Stands in for what the reference gets from Gazebo in its experiments (a poster 0.61 m in front of the camera,
vitvs_v2.py:1391; RGB + uint16 millimetre depth from the RealSense plugin, RealSensePlugin.cpp:250-262; the twist
applied to the camera link, vitvs_v2.py:661-690; the pose read back through tf, :692-700), so that the ROS-free
``ServoLoop`` + ``servo.Controller`` + ``Engine`` can be driven in closed loop by tests/test_gpu_loop.py.

Conventions
  * world frame = the goal camera's optical frame (x right, y down, z forward); the plane is z = plane_z.
  * camera pose (R, t): X_world = R @ X_cam + t.
  * the controller's v_c = (v, w) is the camera's velocity expressed in its OWN optical frame (the IBVS convention
    s' = L v_c); the simulator integrates it as a body twist: t += R v dt, R = R expm([w]x dt).
  * ``apply_twist(lin, ang)`` receives what ``publish_twist`` would publish: the Gazebo axis remap
    lin = (v2, -v0, -v1), ang = (w2, -w0, -w1) (vitvs_v2.py:671-676); it is undone here.
Rendering is a homography: every camera pixel's ray is intersected with the plane and the texture is sampled
bilinearly (torch.grid_sample; on the GPU when a device is given — plumbing for the test, not product arithmetic).
"""
from __future__ import annotations

import numpy as np
import torch


def rodrigues(w: np.ndarray) -> np.ndarray:
    """Rotation matrix exp([w]x) of a rotation vector."""
    th = float(np.linalg.norm(w))
    K = np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + (np.sin(th) / th) * K + ((1.0 - np.cos(th)) / (th * th)) * (K @ K)


def quat_xyzw(R: np.ndarray) -> np.ndarray:
    """Unit quaternion (x, y, z, w) of a rotation matrix."""
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax([R[0, 0], R[1, 1], R[2, 2]]))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    return q / np.linalg.norm(q)


class PlanarScene:
    def __init__(self, texture_u8: np.ndarray, metres_per_px: float, params, plane_z: float = 0.61, device="cpu"):
        self.dev = torch.device(device)
        self.tex = torch.from_numpy(np.ascontiguousarray(texture_u8)).to(self.dev).permute(2, 0, 1)[None].to(torch.float32)
        self.th, self.tw = texture_u8.shape[:2]
        self.mpp = float(metres_per_px)
        self.plane_z = float(plane_z)
        self.p = params
        v, u = torch.meshgrid(torch.arange(params.v_max, dtype=torch.float64), torch.arange(params.u_max, dtype=torch.float64),
                              indexing="ij")
        self.rays = torch.stack([(u - params.c_x) / params.f_x, (v - params.c_y) / params.f_y, torch.ones_like(u)], -1).to(self.dev)

    def render(self, R: np.ndarray, t: np.ndarray):
        """-> (rgb uint8 [v_max, u_max, 3], depth uint16 millimetres [v_max, u_max]) as numpy arrays."""
        Rt = torch.from_numpy(np.asarray(R, np.float64)).to(self.dev)
        tt = torch.from_numpy(np.asarray(t, np.float64)).to(self.dev)
        rw = self.rays @ Rt.T                                     # ray directions in the world frame
        s = (self.plane_z - tt[2]) / rw[..., 2]                   # depth along the optical axis (the ray's z_cam is 1)
        X = tt[0] + s * rw[..., 0]
        Y = tt[1] + s * rw[..., 1]
        px = X / self.mpp + (self.tw - 1) / 2.0                   # texture pixel coordinates
        py = Y / self.mpp + (self.th - 1) / 2.0
        grid = torch.stack([(2 * px + 1) / self.tw - 1, (2 * py + 1) / self.th - 1], -1)[None].to(torch.float32)
        img = torch.nn.functional.grid_sample(self.tex, grid, mode="bilinear", padding_mode="border", align_corners=False)
        rgb = img[0].permute(1, 2, 0).round().clamp(0, 255).to(torch.uint8)
        depth = (s * 1000.0).round().clamp(0, 65535).to(torch.int32)
        return rgb.cpu().numpy(), depth.cpu().numpy().astype(np.uint16)


class CameraSim:
    """Pose state + the three callables ``ServoLoop`` wants (get_pose, apply_twist, sense)."""

    def __init__(self, scene: PlanarScene, controller, R0: np.ndarray, t0: np.ndarray, dt: float):
        self.scene, self.ctl, self.dt = scene, controller, float(dt)
        self.R, self.t = np.array(R0, np.float64), np.array(t0, np.float64)
        self.frames = 0
        self.last_rgb = self.last_depth = None

    def get_pose(self):
        return self.t.copy(), quat_xyzw(self.R)

    def apply_twist(self, lin, ang):
        v = np.array([-lin[1], -lin[2], lin[0]])                  # undo the Gazebo remap (vitvs_v2.py:671-676)
        w = np.array([-ang[1], -ang[2], ang[0]])
        self.t = self.t + self.R @ v * self.dt
        self.R = self.R @ rodrigues(w * self.dt)

    def sense(self):
        self.last_rgb, self.last_depth = self.scene.render(self.R, self.t)
        self.ctl.image_callback_rgb(self.last_rgb)
        self.ctl.image_callback_depth(self.last_depth)
        self.frames += 1
