"""ROS-free servo loop around the hot path: the step after ``compute_velocity`` in the reference.

Mirrors, without rospy / Gazebo (reference paths under catkin_ws/ibvs/src/vitvs_v2.py):

  ServoLoop.run()                      Controller.run                      :702-819
  ServoLoop.is_visual_servoing_done()  Controller.is_visual_servoing_done  :345-421
  ServoLoop.calculate_end_error()      Controller.calculate_end_error      :843-861
  ServoRunResult                       the 19-tuple ``run`` returns        :801-819 (same order, so result[i] ports 1:1)

What the reference gets from ROS is injected as callables:
  get_pose()            -> (position[3] in metres, quaternion[4] xyzw)     get_current_camera_pose (tf lookup)
  apply_twist(lin, ang)    called with what publish_twist would publish    self.pub.publish(Twist)
  sense()                  optional; called at the top of every iteration to feed the controller new frames
                           (image_callback_rgb / image_callback_depth), the rospy subscriber threads' job
  is_shutdown()            optional; rospy.is_shutdown

``controller`` is anything with the reference Controller's hot-path surface: ``ibvs()``, ``v_c``,
``velocity_vector_history``, ``latest_image``, ``publish_twist(v_c) -> (lin, ang)`` — i.e. ``servo.Controller``.
The thresholds are the reference's literals (300-iteration floor, 2x divergence abort, 90 % / 1 cm / 1 degree error
tests, two 100-sample velocity windows with 1 mm/s and 0.1 deg/s), kept as class attributes.  Host-side Python, as in
the reference; pinned by tests/golden/servo_loop.npz, generated from the reference's own methods
(oracle/make_loop_golden.py).
"""
from __future__ import annotations

from typing import Callable, NamedTuple, Optional

import numpy as np


class ServoRunResult(NamedTuple):
    """Field order = the reference's return tuple (vitvs_v2.py:801-819)."""
    camera_position: object
    orientation_quaternion: object
    converged: bool
    position_error: float            # cm
    orientation_error: float         # degrees
    position_history: np.ndarray
    orientation_history: np.ndarray
    iteration_count: int
    lowest_position_error: float
    lowest_orientation_error: float
    average_velocities: np.ndarray
    velocity_mean_100: np.ndarray
    velocity_mean_10: np.ndarray
    applied_velocity_x: np.ndarray
    applied_velocity_y: np.ndarray
    applied_velocity_z: np.ndarray
    applied_velocity_roll: np.ndarray
    applied_velocity_pitch: np.ndarray
    applied_velocity_yaw: np.ndarray


def rotation_angle_deg(q_current, q_desired) -> float:
    """Angle of the rotation taking ``q_current`` to ``q_desired`` (xyzw quaternions), in degrees:
    scipy's ``(R.from_quat(cur).inv() * R.from_quat(des)).magnitude() * 180 / pi`` (vitvs_v2.py:857-859)."""
    a = np.asarray(q_current, dtype=np.float64)
    b = np.asarray(q_desired, dtype=np.float64)
    a = a / np.linalg.norm(a)
    b = b / np.linalg.norm(b)
    ax, ay, az, aw = -a[0], -a[1], -a[2], a[3]          # inverse of a unit quaternion
    bx, by, bz, bw = b
    x = aw * bx + ax * bw + ay * bz - az * by
    y = aw * by - ax * bz + ay * bw + az * bx
    z = aw * bz + ax * by - ay * bx + az * bw
    w = aw * bw - ax * bx - ay * by - az * bz
    return float(2.0 * np.arctan2(np.sqrt(x * x + y * y + z * z), abs(w)) * (180 / np.pi))


class ServoLoop:
    MIN_ITERATIONS = 300                 # :347
    DIVERGENCE_FACTOR = 2                # :359
    ERROR_RATIO = 0.1                    # :364-365 (90 % reduction)
    ERROR_ABS_TRANSLATION_CM = 0.01      # :367 (the reference compares centimetres with 0.01 and calls it 1 cm)
    ERROR_ABS_ROTATION_DEG = 1.0         # :368
    WINDOW = 100                         # :374-379, two windows
    WINDOW_TRANS_MM_S = 1.0              # :401
    WINDOW_ROT_DEG_S = 0.1               # :401

    def __init__(self, controller, desired_position, desired_orientation, *, get_pose: Callable,
                 apply_twist: Optional[Callable] = None, sense: Optional[Callable] = None,
                 is_shutdown: Optional[Callable] = None, max_iterations: int = 1500, log: Optional[Callable] = None):
        self.controller = controller
        self.desired_position = np.asarray(desired_position, dtype=np.float64)
        self.desired_orientation = np.asarray(desired_orientation, dtype=np.float64)
        self.get_pose = get_pose
        self.apply_twist = apply_twist
        self.sense = sense
        self.is_shutdown = is_shutdown
        self.max_iterations = int(max_iterations)     # config.yaml:34
        self.log = log or (lambda msg: None)
        self.camera_position = None
        self.orientation_quaternion = None
        self.iteration_count = 0
        self.initial_error_translation = None
        self.initial_error_rotation = None
        self._reset_histories()

    def _reset_histories(self):
        self.position_history = []
        self.orientation_history = []
        self.velocity_history = []
        self.average_velocities = []
        self.velocity_mean_100 = []
        self.velocity_mean_10 = []
        self.applied_velocity_x = []
        self.applied_velocity_y = []
        self.applied_velocity_z = []
        self.applied_velocity_roll = []
        self.applied_velocity_pitch = []
        self.applied_velocity_yaw = []

    # ------------------------------------------------------------------ errors
    def calculate_end_error(self):
        """(position error in cm, orientation error in degrees) of the current pose against the desired one."""
        position_error = np.linalg.norm(np.asarray(self.camera_position) - self.desired_position) * 100
        return position_error, rotation_angle_deg(self.orientation_quaternion, self.desired_orientation)

    # ------------------------------------------------------------------ convergence monitor
    def is_visual_servoing_done(self):
        """→ (done, converged), the reference's rules in the reference's order."""
        if self.iteration_count < self.MIN_ITERATIONS:
            return False, False
        err_t, err_r = self.calculate_end_error()
        if err_t > self.DIVERGENCE_FACTOR * self.initial_error_translation:
            self.log("Aborting sample due to position error exceeding twice the initial error.")
            return True, False
        reduced_90 = bool((err_t / self.initial_error_translation) < self.ERROR_RATIO and
                          (err_r / self.initial_error_rotation) < self.ERROR_RATIO)
        # (the reference also forms "error below 1 cm and 1 degree" here and never reads it)
        hist = self.controller.velocity_vector_history
        if len(hist) >= 2 * self.WINDOW:
            recent = np.array(hist[-2 * self.WINDOW:])
            first, second = recent[:self.WINDOW], recent[self.WINDOW:]
            first_trans = np.mean(np.linalg.norm(first[:, :3] * 1000.0, axis=1))        # mm/s
            first_rot = np.mean(np.linalg.norm(np.degrees(first[:, 3:]), axis=1))      # deg/s
            second_trans = np.mean(np.linalg.norm(second[:, :3] * 1000.0, axis=1))
            second_rot = np.mean(np.linalg.norm(np.degrees(second[:, 3:]), axis=1))
            if first_trans < self.WINDOW_TRANS_MM_S and first_rot < self.WINDOW_ROT_DEG_S:
                if second_trans > first_trans and second_rot > first_rot:
                    self.log("Velocity trend indicates convergence - checking final error")
                    return True, reduced_90
        if self.iteration_count >= self.max_iterations:
            self.log("Maximum iterations reached")
            return True, reduced_90
        return False, False

    # ------------------------------------------------------------------ the loop
    def _publish(self, v_c):
        lin, ang = self.controller.publish_twist(v_c)
        self.applied_velocity_x.append(lin[0])
        self.applied_velocity_y.append(lin[1])
        self.applied_velocity_z.append(lin[2])
        self.applied_velocity_roll.append(ang[0])
        self.applied_velocity_pitch.append(ang[1])
        self.applied_velocity_yaw.append(ang[2])
        if self.apply_twist is not None:
            self.apply_twist(lin, ang)

    def _result(self, converged, pos_err, rot_err, low_pos, low_rot, histories=True):
        arr = (lambda x: np.array(x)) if histories else (lambda x: np.array([]))
        return ServoRunResult(self.camera_position, self.orientation_quaternion, converged, pos_err, rot_err,
                              arr(self.position_history), arr(self.orientation_history),
                              self.iteration_count if histories else 0, low_pos, low_rot,
                              arr(self.average_velocities), arr(self.velocity_mean_100), arr(self.velocity_mean_10),
                              arr(self.applied_velocity_x), arr(self.applied_velocity_y), arr(self.applied_velocity_z),
                              arr(self.applied_velocity_roll), arr(self.applied_velocity_pitch),
                              arr(self.applied_velocity_yaw))

    def run(self):
        """One servoing episode → ``ServoRunResult`` (``None`` if the initial pose is unavailable, like the reference).

        Per iteration: ``controller.ibvs()`` (a persistent feature failure ends the episode unconverged with empty
        histories), bookkeeping of the mean |v_c| and its 100 / 10-sample running means, the twist is applied, the new
        pose is read, errors are tracked, and ``is_visual_servoing_done`` decides.  Any other exception ends the
        episode with the histories gathered so far and infinite errors (the reference's error tuple)."""
        ctl = self.controller
        self.iteration_count = 0
        ctl.velocity_vector_history = []
        ctl.feature_failure_count = 0
        self._reset_histories()
        self.camera_position, self.orientation_quaternion = self.get_pose()
        if self.camera_position is None or self.orientation_quaternion is None:
            self.log("Failed to get initial camera pose")
            return None
        if self.initial_error_translation is None:
            self.initial_error_translation, self.initial_error_rotation = self.calculate_end_error()
        lowest_pos, lowest_rot = float("inf"), float("inf")
        try:
            while not (self.is_shutdown is not None and self.is_shutdown()):
                if self.sense is not None:
                    self.sense()
                if ctl.latest_image is None:
                    continue
                try:
                    ctl.ibvs()
                except RuntimeError as exc:
                    if str(exc) == "Persistent feature detection failure":
                        self.log("Aborting sample due to persistent feature detection failures")
                        return self._result(False, float("inf"), float("inf"), float("inf"), float("inf"),
                                            histories=False)
                    raise
                self.iteration_count += 1
                avg = np.mean(np.abs(ctl.v_c))
                self.average_velocities.append(avg)
                self.velocity_history.append(avg)
                self.velocity_mean_100.append(np.mean(self.velocity_history[-100:]) if len(self.velocity_history) >= 100
                                              else np.mean(self.velocity_history))
                self.velocity_mean_10.append(np.mean(self.velocity_history[-10:]) if len(self.velocity_history) >= 10
                                             else np.mean(self.velocity_history))
                self._publish(ctl.v_c)
                self.camera_position, self.orientation_quaternion = self.get_pose()
                self.position_history.append(self.camera_position)
                self.orientation_history.append(self.orientation_quaternion)
                pos_err, rot_err = self.calculate_end_error()
                lowest_pos, lowest_rot = min(lowest_pos, pos_err), min(lowest_rot, rot_err)
                done, converged = self.is_visual_servoing_done()
                if done:
                    return self._result(converged, pos_err, rot_err, lowest_pos, lowest_rot)
        except Exception as exc:  # noqa: BLE001  (the reference catches everything here, vitvs_v2.py:821-823)
            self.log(f"Error in run loop: {exc}")
            return self._result(False, float("inf"), float("inf"), float("inf"), float("inf"))
        return None
