"""ORACLE tooling (build container only): fixtures for the servo loop and its convergence monitor.

    python oracle/make_loop_golden.py            # writes tests/golden/servo_loop.npz

Runs the REFERENCE's own ``Controller.run``, ``is_visual_servoing_done``, ``calculate_end_error``,
``publish_twist`` and ``_create_error_return_tuple`` (vitvs_v2.py:702-841, 345-421, 843-861, 661-690), loaded from
/root/reference by AST (oracle/ref_extract.py: no module import, rospy is a do-nothing object, ``Twist`` a plain
record, the publisher a list), on SCRIPTED episodes: per-iteration twists ``v_c[k]`` and camera poses ``pose[k]``
fixed in advance, so that the loop's bookkeeping and every exit rule is exercised without a simulator.  The
fixture is data only: the scripts and the 19-tuple the reference returned.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
from scipy.spatial.transform import Rotation as R

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_extract as rx  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
FIELDS = ("camera_position", "orientation_quaternion", "converged", "position_error", "orientation_error",
          "position_history", "orientation_history", "iteration_count", "lowest_position_error",
          "lowest_orientation_error", "average_velocities", "velocity_mean_100", "velocity_mean_10",
          "applied_velocity_x", "applied_velocity_y", "applied_velocity_z", "applied_velocity_roll",
          "applied_velocity_pitch", "applied_velocity_yaw")


class _Vec:
    x = y = z = 0.0


class _Twist:
    def __init__(self):
        self.linear, self.angular = _Vec(), _Vec()


def reference_loop(script_v, script_pose, desired_position, desired_orientation, *, fail_from=None, max_iterations=1500,
                   max_velocity=1.0, first_v_none=False):
    """The reference Controller's loop methods bound to a scripted episode."""
    methods = ("run", "is_visual_servoing_done", "calculate_end_error", "publish_twist", "_create_error_return_tuple")
    ns = rx._compile_defs(rx.VITVS_PY, class_methods={"Controller": methods},
                          namespace={"np": np, "rospy": rx._QuietLog(), "R": R, "Twist": _Twist})
    ctl = ns["Controller"]()
    published = []
    ctl.pub = types.SimpleNamespace(publish=published.append)
    ctl.desired_position = np.asarray(desired_position, float)
    ctl.desired_orientation = np.asarray(desired_orientation, float)
    ctl.max_velocity, ctl.max_iterations, ctl.max_velocity_vector_history = max_velocity, max_iterations, 200
    ctl.latest_image = object()
    ctl.initial_error_translation = ctl.initial_error_rotation = None
    ctl.v_c = None
    ctl.camera_position = ctl.orientation_quaternion = None
    ctl.iteration_count = 0
    calls = dict(pose=0, ibvs=0)

    def get_current_camera_pose():
        k = min(calls["pose"], len(script_pose) - 1)
        calls["pose"] += 1
        return script_pose[k, :3].copy(), script_pose[k, 3:].copy()

    def ibvs():                                          # stands in for the hot path: a scripted twist per iteration
        k = calls["ibvs"]
        calls["ibvs"] += 1
        if fail_from is not None and k >= fail_from:
            raise RuntimeError("Persistent feature detection failure")
        if first_v_none and k == 0:
            return                                       # a skipped first update leaves v_c = None (vitvs_v2.py:224)
        ctl.v_c = script_v[min(k, len(script_v) - 1)].copy()
        ctl.velocity_vector_history.append(ctl.v_c)      # what the real ibvs does, :626-628
        if len(ctl.velocity_vector_history) > ctl.max_velocity_vector_history:
            ctl.velocity_vector_history.pop(0)

    ctl.get_current_camera_pose = get_current_camera_pose
    ctl.ibvs = ibvs
    return ctl.run()


def _quat(rotvec):
    return R.from_rotvec(rotvec).as_quat()


def make_scripts():
    """name -> kwargs of reference_loop.  Poses: position error decays / grows as scripted, orientation likewise."""
    rng = np.random.default_rng(20250706)
    des_p = np.array([0.0, 0.0, 0.61])
    des_q = _quat([0.0, 0.0, 0.0])
    cases = {}

    def episode(n, pos_err0, rot0_deg, decay, v_scale, v_decay, grow_after=None, noise=0.0):
        k = np.arange(n + 1)[:, None]
        perr = np.asarray(pos_err0)[None] * np.exp(-decay * k)
        axis = np.array([0.3, -0.5, 0.8]) / np.linalg.norm([0.3, -0.5, 0.8])
        ang = np.radians(rot0_deg) * np.exp(-decay * k[:, 0])
        pose = np.concatenate([des_p[None] + perr, np.stack([_quat(axis * a) for a in ang])], axis=1)
        v = v_scale * np.exp(-v_decay * np.arange(n))[:, None] * rng.uniform(0.5, 1.0, size=(n, 6))
        if grow_after is not None:                       # twists creep up again after `grow_after` iterations
            g = np.maximum(np.arange(n) - grow_after, 0)[:, None]
            v = v * np.exp(0.03 * g)
        v = v + noise * rng.standard_normal((n, 6))
        return v, pose

    # 1. the velocity windows end the episode (small, then rising twists) with the error reduced by > 90 %
    v, pose = episode(700, [0.05, -0.04, 0.08], 20.0, 0.02, 2e-4, 0.01, grow_after=420)
    cases["velocity_trend_converged"] = dict(script_v=v, script_pose=pose, desired_position=des_p, desired_orientation=des_q)
    # 2. the same twists, but the pose error stalls at 40 %: done, not converged
    v2, pose2 = episode(700, [0.05, -0.04, 0.08], 20.0, 0.0013, 2e-4, 0.01, grow_after=420)
    cases["velocity_trend_not_converged"] = dict(script_v=v2, script_pose=pose2, desired_position=des_p,
                                                 desired_orientation=des_q)
    # 3. divergence: the position error passes twice its initial value after the 300-iteration floor
    v3, pose3 = episode(400, [0.03, 0.02, -0.01], 5.0, -0.004, 5e-3, 0.0)
    cases["diverged"] = dict(script_v=v3, script_pose=pose3, desired_position=des_p, desired_orientation=des_q)
    # 4. iteration cap reached with / without the 90 % reduction (large twists keep the windows from firing)
    v4, pose4 = episode(360, [0.05, 0.05, 0.05], 10.0, 0.02, 5e-2, 0.0)
    cases["max_iterations_converged"] = dict(script_v=v4, script_pose=pose4, desired_position=des_p,
                                             desired_orientation=des_q, max_iterations=350)
    v5, pose5 = episode(360, [0.05, 0.05, 0.05], 10.0, 0.002, 5e-2, 0.0)
    cases["max_iterations_not_converged"] = dict(script_v=v5, script_pose=pose5, desired_position=des_p,
                                                 desired_orientation=des_q, max_iterations=350)
    # 5. the hot path reports a persistent feature failure in iteration 37
    cases["persistent_failure"] = dict(script_v=v4, script_pose=pose4, desired_position=des_p, desired_orientation=des_q,
                                       fail_from=37)
    # 6. twists beyond max_velocity are clipped in what is applied, not in the histories of |v_c|
    v6, pose6 = episode(320, [0.05, 0.05, 0.05], 10.0, 0.05, 3.0, 0.02)
    cases["clipped_twists"] = dict(script_v=v6, script_pose=pose6, desired_position=des_p, desired_orientation=des_q,
                                   max_iterations=310, max_velocity=1.0)
    # 7. the first update is skipped (no features yet): v_c is still None, the reference's loop ends in its error tuple
    cases["first_update_skipped"] = dict(script_v=v4, script_pose=pose4, desired_position=des_p, desired_orientation=des_q,
                                         first_v_none=True)
    return cases


def main():
    assert rx.available(), "reference tree not found"
    os.makedirs(GOLDEN, exist_ok=True)
    blob = {}
    for name, kw in make_scripts().items():
        out = reference_loop(**kw)
        assert out is not None and len(out) == len(FIELDS)
        used = (kw["fail_from"] if kw.get("fail_from") is not None else int(out[7])) + 2   # rows the episode touched (+ margin)
        blob[f"{name}/script_v"] = np.asarray(kw["script_v"][:used], dtype=np.float64)
        blob[f"{name}/script_pose"] = np.asarray(kw["script_pose"][:used + 1], dtype=np.float64)
        for k in ("desired_position", "desired_orientation"):
            blob[f"{name}/{k}"] = np.asarray(kw[k], dtype=np.float64)
        blob[f"{name}/fail_from"] = np.int64(-1 if kw.get("fail_from") is None else kw["fail_from"])
        blob[f"{name}/max_iterations"] = np.int64(kw.get("max_iterations", 1500))
        blob[f"{name}/max_velocity"] = np.float64(kw.get("max_velocity", 1.0))
        blob[f"{name}/first_v_none"] = np.bool_(kw.get("first_v_none", False))
        for field, val in zip(FIELDS, out):
            blob[f"{name}/out/{field}"] = np.asarray(val)
        print(f"{name:32s} iterations {int(out[7]):4d} converged {bool(out[2])!s:5s} "
              f"errors {float(out[3]):.4f} cm {float(out[4]):.4f} deg")
    path = os.path.join(GOLDEN, "servo_loop.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
