"""Host-side mirror of the reference's interface around the HIP hot path.

Same names, argument meaning and error behaviour as the reference's callables, without ROS:

  compute_velocity(I_cur, I_des, Z, K)        the north-star seam = detect_features() + ibvs() up to the raw
                                              twist (reference: vitvs_v2.py:464-523, 588-622)
  find_correspondences_batch(desc1, desc2)    reference: vitvs_v2.py:72-155 (same return shapes, same RNG stream)
  Controller.detect_features() / .ibvs()      reference: vitvs_v2.py:464-523, 588-632 (EMA :325-343, failure
                                              counter :500-505, twist remap :661-676)
  Controller.best_rotation(frames)            reference: find_and_set_best_pose, vitvs_v2.py:1151-1189
  ServoLoop (vit-vs_amd/loop.py)              reference: Controller.run / is_visual_servoing_done, :702-819, :345-421

All arithmetic of the path runs on the GPU through ``Engine`` (C ABI); what stays on the host is what the
reference also does on the host: PIL resize, the RNG draw, the EMA state and the twist remap.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from .config import ServoParams
from .engine import Engine, VitvsError

STATUS_NAMES = {0: "ok", 1: "no_correspondence", 2: "too_few_features", 3: "no_depth"}


# ------------------------------------------------------------------------------------------ functional API
def compute_velocity(engine: Engine, I_cur, I_des, Z, K=None, *, selection="order",
                     generator: Optional[torch.Generator] = None, num_pairs: Optional[int] = None):
    """One servo update for one frame pair: raw (pre-EMA) ``v_c`` float64[6] and a status int.

    ``I_cur`` / ``I_des``: uint8 RGB frames already at the extractor's input size (the reference resizes with
    PIL before the path); ``Z``: the sensor's uint16 millimetre depth image; ``K`` = (fx, fy, cx, cy), default
    from the engine's parameters.  ``selection``:
      "order"      a fresh random visiting order (CPU generator); the first num_pairs mutual-NN tokens met are used,
                   everything stays on the device (one graph replay)
      "reference"  the reference's exact procedure and RNG stream (torch sort + randperm on the host between the
                   correspondence and the control law; two device round trips, as in the reference)
      "dense"      every mutual-NN token
      array-like   explicit token ids of the desired frame
    ``num_pairs``: feature pairs of the law for this call (the reference's ``Controller.num_pairs``; default: the
    engine's parameters).
    """
    one = lambda a: a[None] if torch.is_tensor(a) else np.asarray(a)[None]   # noqa: E731  (device tensors stay there)
    des = None if I_des is None else one(I_des)            # None: the goal cached by Engine.set_goal
    v, st = compute_velocity_batch(engine, one(I_cur), des, None if Z is None else one(Z), K, selection=selection,
                                   generator=generator, num_pairs=num_pairs)
    return v[0], int(st[0])


def compute_velocity_batch(engine: Engine, I_cur, I_des, Z, K=None, *, selection="order", des_shared: bool = False,
                           generator: Optional[torch.Generator] = None, num_pairs: Optional[int] = None):
    """``B`` pairs in one call → (v_c float64 [B,6], status int32 [B]) as numpy arrays."""
    p = engine.params
    K = p.intrinsics() if K is None else K
    n = int(I_cur.shape[0]) if torch.is_tensor(I_cur) else np.asarray(I_cur).shape[0]
    if isinstance(selection, str) and selection == "reference":
        if n != 1:
            raise ValueError('selection="reference" follows the reference: one pair per call')
        return _reference_update(engine, I_cur, I_des, Z, K, generator, num_pairs)
    if isinstance(selection, str) and selection == "order":
        order = torch.stack([torch.randperm(engine.tokens, generator=generator) for _ in range(n)]).to(torch.int32)
        v, st = engine.compute_velocity(I_cur, I_des, Z, K, mode=_lib.SELECT_ORDER, selection=order,
                                        des_shared=des_shared, num_pairs=num_pairs)
    elif isinstance(selection, str) and selection == "dense":
        v, st = engine.compute_velocity(I_cur, I_des, Z, K, mode=_lib.SELECT_DENSE, des_shared=des_shared)
    else:
        ids = selection if isinstance(selection, (list, tuple)) and n > 1 else [selection]
        v, st = engine.compute_velocity(I_cur, I_des, Z, K, mode=_lib.SELECT_EXPLICIT, selection=list(ids),
                                        des_shared=des_shared, num_pairs=num_pairs)
    return v.cpu().numpy(), st.cpu().numpy()


def _candidate_order(nn_1: torch.Tensor, nn_2: torch.Tensor, grid: int, distance_threshold: float = 1.0):
    """Tokens that pass the reference's cyclic-consistency filter, in the order its descending sort leaves them
    (so that a following ``randperm`` picks the same tokens as the reference does with the same seed)."""
    t = nn_1.numel()
    back = nn_2[nn_1]                                    # where each token's best match points back to
    here = torch.arange(t)
    delta = torch.stack((back // grid - here // grid, back % grid - here % grid), dim=-1)
    dist = -torch.linalg.vector_norm((delta + 1e-6).to(torch.float32), 2, dim=-1)
    spread = dist - dist.min()
    spread = spread / (spread.max() + 1e-8)
    vals, order = spread.sort(dim=-1, descending=True)
    return order[vals >= distance_threshold]


def find_correspondences_batch(engine: Engine, descriptors1: torch.Tensor, descriptors2: torch.Tensor,
                               num_pairs: int = 18, distance_threshold: float = 1):
    """Drop-in for the reference function: descriptors ``[1,1,T,D']`` → ``(points1, points2, sim)`` with
    ``points*`` int64 ``[K,2]`` (row, col) and ``sim`` ``[1,K]``, or ``(None, None, None)``.
    Similarities and arg-maxes run on the GPU; the draw uses torch's global CPU RNG like the reference."""
    d1 = descriptors1.reshape(-1, descriptors1.shape[-1])
    d2 = descriptors2.reshape(-1, descriptors2.shape[-1])
    t = d1.shape[0]
    grid = int(np.sqrt(t))
    nn_1, nn_2, sim_1 = (x.cpu() for x in engine.correspond(d1, d2))
    nn_1, nn_2 = nn_1.long(), nn_2.long()
    rc = lambda idx: torch.stack((idx // grid, idx % grid), dim=-1)  # noqa: E731
    if sim_1.mean().item() > 0.99:                       # same-image shortcut
        k = min(num_pairs, t)
        pts = rc(torch.randperm(t)[:k])
        return pts, pts.clone(), torch.ones(k)
    cand = _candidate_order(nn_1, nn_2, grid, distance_threshold)
    k = min(num_pairs, cand.numel())
    if k == 0:
        return None, None, None
    chosen = cand[torch.randperm(cand.numel())[:k]]
    return rc(chosen), rc(nn_1[chosen]), sim_1[chosen].unsqueeze(0)


def _reference_update(engine: Engine, I_cur, I_des, Z, K, generator, num_pairs=None):
    """The reference's own sequence with ONE forward: similarity + arg-max on the device, the reference's sort + ``randperm``
    draw on the host (same torch RNG stream), then the control law on the device for the drawn tokens.

    Host arrays in the engine's frame geometry (what the ``Controller``'s callbacks hold) take the two-call form: the host-pointer
    update (``vitvs_compute_velocity``; its own law, on a fixed visiting order, is thrown away) leaves the arg-max tables in host
    memory, the draw runs, and ``vitvs_reselect`` evaluates the law for the drawn tokens on the keys, depth image and intrinsics
    the first call left in the handle.  Device tensors (``Controller._resized``) and other geometries go through the descriptor
    seams (``extract_descriptors`` -> ``correspond`` -> ``servo_from_nn``), as before."""
    k = engine.params.num_pairs if num_pairs is None else int(num_pairs)

    def draw(nn_1, nn_2, sim_1):
        return _draw_like_the_reference(nn_1, nn_2, sim_1, engine.cfg.grid, k, generator)

    host = (not torch.is_tensor(I_cur) and not torch.is_tensor(I_des) and I_des is not None
            and tuple(np.asarray(I_cur).shape[-3:-1]) == tuple(engine.frame_size)
            and (Z is None or (not torch.is_tensor(Z) and np.asarray(Z).dtype == np.uint16)))
    if host:
        order = np.arange(engine.tokens, dtype=np.int32)
        _, st0 = engine.compute_velocity_host(I_cur, I_des, Z, K, _lib.SELECT_ORDER, order, num_pairs=k)
        tab = engine.last_tables(1)
        ids = draw(tab["nn_1"][0], tab["nn_2"][0], tab["sim_1"][0])
        if ids is None:                                   # (None, None, None) in the reference
            return np.zeros((1, 6)), np.array([_lib.STATUS_NO_CORRESPONDENCE], np.int32)
        v, st = engine.reselect_host(_lib.SELECT_EXPLICIT, [ids.to(torch.int32).numpy()], num_pairs=k)
        return v, st
    both = torch.cat([torch.as_tensor(I_des).to(engine.device), torch.as_tensor(I_cur).to(engine.device)])
    desc = engine.extract_descriptors(both)
    d1, d2 = desc[0, 0], desc[1, 0]
    nn_1, nn_2, sim_1 = engine.correspond(d1, d2)
    ids = draw(nn_1.cpu().long(), nn_2.cpu().long(), sim_1.cpu())
    z = None if Z is None else torch.as_tensor(Z).reshape(engine.params.v_max, engine.params.u_max)
    if ids is None:                                       # (None, None, None) in the reference
        return np.zeros((1, 6)), np.array([_lib.STATUS_NO_CORRESPONDENCE], np.int32)
    v, st = engine.servo_from_nn(nn_1, nn_2, sim_1, z, K, mode=_lib.SELECT_EXPLICIT, selection=[ids.to(torch.int32)],
                                 num_pairs=k)
    return v.cpu().numpy()[None], np.array([int(st)], np.int32)


_GRID_CACHE: dict = {}


def _candidate_order_host(nn_1: np.ndarray, nn_2: np.ndarray, grid: int) -> torch.Tensor:
    """``_candidate_order`` on host arrays: the integer part (index arithmetic) in numpy, every floating-point step — the shifted
    norm, the normalised spread, the descending sort whose tie order the draw inherits — through the same torch calls on the same
    values, so the result is the same tensor (tests/test_servo_host.py compares the two over random and degenerate tables).  The
    ``Controller``'s default selection runs this between two device calls on every update: ~15 small torch ops were 0.1 ms of it."""
    t = int(nn_1.shape[0])
    here = _GRID_CACHE.get((t, grid))
    if here is None:
        idx = np.arange(t, dtype=np.int64)
        here = _GRID_CACHE[(t, grid)] = (idx // grid, idx % grid)
    back = nn_2[nn_1].astype(np.int64, copy=False)
    delta = np.empty((t, 2), np.int64)
    np.subtract(back // grid, here[0], out=delta[:, 0])
    np.subtract(back % grid, here[1], out=delta[:, 1])
    dist = -torch.linalg.vector_norm((torch.from_numpy(delta) + 1e-6).to(torch.float32), 2, dim=-1)
    spread = dist - dist.min()
    spread = spread / (spread.max() + 1e-8)
    vals, order = spread.sort(dim=-1, descending=True)
    return order[vals >= 1.0]


def _draw_like_the_reference(nn_1, nn_2, sim_1, grid: int, num_pairs: int, generator=None):
    """Token ids of the desired frame as ``find_correspondences_batch`` draws them (vitvs_v2.py:84-141), or None.  ``generator``:
    the torch CPU generator standing in for the reference's global RNG (``torch.randperm(n, generator=g)`` draws what
    ``torch.randperm(n)`` draws from a global RNG in the same state; None: the global RNG itself).  Tensors or host arrays."""
    if not torch.is_tensor(nn_1):
        nn_1, nn_2, sim_1 = np.asarray(nn_1), np.asarray(nn_2), np.asarray(sim_1)
        t = int(nn_1.shape[0])
        if float(torch.from_numpy(sim_1).mean()) > 0.99:     # same-image shortcut (torch's fp32 mean, like the reference's)
            return torch.randperm(t, generator=generator)[:min(num_pairs, t)]
        cand = _candidate_order_host(nn_1, nn_2, grid)
    else:
        t = nn_1.numel()
        if sim_1.mean().item() > 0.99:                       # same-image shortcut
            return torch.randperm(t, generator=generator)[:min(num_pairs, t)]
        cand = _candidate_order(nn_1, nn_2, grid)
    k = min(num_pairs, cand.numel())
    if k == 0:
        return None
    return cand[torch.randperm(cand.numel(), generator=generator)[:k]]


def ema_update(state: list, v: Sequence[float], alpha: float) -> np.ndarray:
    """Per-component exponential moving average with the reference's first-sample initialisation."""
    out = np.empty(6)
    for i, x in enumerate(np.asarray(v, dtype=np.float64).reshape(6)):
        state[i] = x if state[i] is None else alpha * x + (1 - alpha) * state[i]
        out[i] = state[i]
    return out


def twist_from_velocity(v_c, max_velocity: float):
    """Camera optical frame → (linear xyz, angular xyz) as the reference publishes them (clipped)."""
    c = lambda x: float(min(max(x, -max_velocity), max_velocity))  # noqa: E731
    return (c(v_c[2]), c(-v_c[0]), c(-v_c[1])), (c(v_c[5]), c(-v_c[3]), c(-v_c[4]))


# ------------------------------------------------------------------------------------------ Controller adapter
class Controller:
    """ROS-free stand-in for the reference ``Controller``'s hot-path half: feed it frames, call ``ibvs()``
    in the control loop, read ``v_c``.  Attribute names follow the reference so its ``run()`` logic ports 1:1."""

    def __init__(self, engine: Engine, goal_image, params: Optional[ServoParams] = None,
                 selection: str = "reference"):
        self.engine = engine
        self.params = params or engine.params
        self.num_pairs = self.params.num_pairs
        self.dino_input_size = engine.cfg.img_size
        self.goal_image = goal_image                      # PIL image or uint8 array, any size
        self.latest_image = None                          # uint8 HxWx3 RGB (the reference keeps BGR + a PIL copy)
        self.latest_pil_image = None
        self.latest_image_depth = None                    # uint16 millimetres, v_max x u_max
        self.selection = selection
        self.feature_failure_count = 0
        self.ema_velocities = [None] * 6
        self.v_c = None                                   # like the reference (vitvs_v2.py:224): unset until the first update
        self.velocity_vector_history = []
        self.max_velocity_vector_history = 200            # config.yaml:37
        self.last_status = None
        self.generator = None                             # torch.Generator of the "order" / "reference" draws (None: torch's global RNG)
        self._goal_key, self._goal_np = None, None        # a private, never-written copy of the goal frame: its address tells the
                                                          # library that the staged goal is still the goal (option "reuse_goal_frames")
        self._rejected_geometries = set()                 # camera geometries the fused resize refused (set_frame_size), per engine

    # -- inputs (the reference's ROS callbacks)
    def image_callback_rgb(self, rgb_u8):
        self.latest_image = np.asarray(rgb_u8)
        self.latest_pil_image = self.latest_image

    def image_callback_depth(self, depth_u16):
        self.latest_image_depth = np.asarray(depth_u16)

    def _resized(self, img):
        """The reference's ``image.resize((S, S))`` (PIL default filter: bicubic, vitvs_v2.py:474-475), done on the
        device by ``Engine.resize_frames`` — bit-identical to PIL, no host round trip.  Accepts a uint8 HxWx3 array or
        a PIL image; returns uint8 [S,S,3] (a device tensor when a resize was needed)."""
        s = self.dino_input_size
        if hasattr(img, "convert") and not isinstance(img, np.ndarray):   # PIL image
            img = np.asarray(img.convert("RGB"), dtype=np.uint8)
        arr = np.asarray(img, dtype=np.uint8)
        if arr.shape[:2] == (s, s):
            return arr
        return self.engine.resize_frames(arr)[0]

    def _path_frames(self, *imgs):
        """The frames as the engine takes them.  When all share one geometry (the usual case: one camera), the frames
        themselves: ``Engine.set_frame_size`` makes the launch that builds the patch rows apply the reference's
        ``image.resize((S, S))`` (vitvs_v2.py:474-475) on the way, bit-identical to PIL and without a resized image in
        memory.  Mixed geometries: each frame resized on its own (``_resized``)."""
        arrs = []
        for img in imgs:
            if hasattr(img, "convert") and not isinstance(img, np.ndarray):   # PIL image
                img = np.asarray(img.convert("RGB"), dtype=np.uint8)
            arrs.append(img if torch.is_tensor(img) else np.asarray(img, dtype=np.uint8))
        shapes = {tuple(a.shape[:2]) for a in arrs}
        if len(shapes) == 1 and next(iter(shapes)) not in self._rejected_geometries:
            shape = shapes.pop()
            try:
                self.engine.set_frame_size(*shape)
                return arrs
            except VitvsError:
                # a geometry the fused resize cannot take (too many camera rows per patch for its LDS rows, or no memory for
                # the staging buffers): the engine kept its previous geometry (vitvs_set_frame_size changes nothing on
                # failure); fall back to the stand-alone resize launch below — and do not ask again for this geometry: every
                # attempt builds Pillow's tables, allocates and synchronises the device
                self._rejected_geometries.add(shape)
        self.engine.set_frame_size()
        return [self._resized(a) for a in arrs]

    # -- the hot path
    def detect_features(self):
        """→ ((s_uv_star, s_uv), sim_selected) with int arrays [num_pairs,2] in camera pixels, or (None, None).
        The 10th consecutive failure raises RuntimeError("Persistent feature detection failure")."""
        if self.latest_image is None:
            return None, None
        cur, des = self._path_frames(self.latest_pil_image, self.goal_image)
        if not torch.is_tensor(des):
            # the goal image does not change from update to update (vitvs_v2.py:264): hand the library the SAME private buffer every
            # time, so that it forwards the goal frame already in device memory instead of staging it again (its tokens are still
            # recomputed on every update, like the reference does)
            key = (id(self.goal_image), des.shape)
            if self._goal_key != key:
                self._goal_key, self._goal_np = key, np.array(des, dtype=np.uint8, copy=True, order="C")
                self.engine.set_option("reuse_goal_frames", 1)
            des = self._goal_np
        depth = self.latest_image_depth
        # the law needs a depth image; detect_features itself does not, so feed a dummy one if it is missing
        z = depth if depth is not None else np.zeros((self.params.v_max, self.params.u_max), np.uint16)
        eng = self.engine
        if isinstance(self.selection, str) and self.selection in ("order", "dense") and not torch.is_tensor(cur) \
                and not torch.is_tensor(des) and tuple(cur.shape[:2]) == tuple(eng.frame_size) and np.asarray(z).dtype == np.uint16:
            # the callbacks' own arrays straight into the host-pointer entry point: one C call, no torch tensor, the feature
            # rows come back with it (Engine.last_features reads them from the handle's pinned block)
            if self.selection == "order":
                order = torch.randperm(eng.tokens, generator=self.generator).to(torch.int32).numpy()
                v, st = eng.compute_velocity_host(cur, des, z, self.params.intrinsics(), _lib.SELECT_ORDER, order,
                                                  num_pairs=self.num_pairs)
            else:
                v, st = eng.compute_velocity_host(cur, des, z, self.params.intrinsics(), _lib.SELECT_DENSE, num_pairs=self.num_pairs)
            if not self._absorb(v[0], int(st[0])):
                return None, None
            return self._features(eng.last_features(1), 0)
        v, st = compute_velocity(eng, cur, des, z, self.params.intrinsics(), selection=self.selection,
                                 generator=self.generator, num_pairs=self.num_pairs)
        if not self._absorb(v, st):
            return None, None
        return self._features(eng.last_features(1), 0)

    def _absorb(self, v, st) -> bool:
        """Bookkeeping of one finished update (whoever computed it: this controller's engine, or a ``MultiController``'s batched
        call / pipeline slot): raw twist, status, the consecutive-failure counter of vitvs_v2.py:500-505."""
        self._raw_v, self.last_status = v, st
        if st == _lib.STATUS_NO_CORRESPONDENCE:
            self.feature_failure_count += 1
            if self.feature_failure_count >= 10:
                raise RuntimeError("Persistent feature detection failure")
            return False
        self.feature_failure_count = 0
        return True

    def _features(self, det, b: int = 0):
        """``detect_features``' return value from pair ``b`` of an engine's ``last_details``."""
        k = self.num_pairs
        s_uv_star = det["s_uv"][b, :k, 0:2].astype(int)
        s_uv = det["s_uv"][b, :k, 2:4].astype(int)
        n_matched = int(det["info"][b, 3])
        sim = torch.from_numpy(det["feat"][b, :max(n_matched, 0), 3].astype(np.float32)).unsqueeze(0)
        return (s_uv_star, s_uv), sim

    def ibvs(self):
        """One control-law step: updates ``self.v_c`` (EMA-smoothed) or leaves it untouched on failure."""
        if self.latest_image is None:
            return
        result = self.detect_features()
        self._law_step(result is not None and result[0] is not None)

    def _law_step(self, have_features: bool):
        if not have_features:
            return
        if self.latest_image_depth is None:               # reference: "Failed to get depth - skipping"
            return
        # fewer than 4 matches: the reference's calculate_uv hands back all-zero feature arrays of num_pairs rows
        # (vitvs_v2.py:539-541), len() >= 4 passes the check at :604, e = 0 and the raw twist is exactly 0 — which is
        # what the kernel reports with status TOO_FEW; the EMA is updated with it, as in the reference
        self.v_c = ema_update(self.ema_velocities, self._raw_v, self.params.ema_alpha)
        self.velocity_vector_history.append(self.v_c)
        if len(self.velocity_vector_history) > self.max_velocity_vector_history:
            self.velocity_vector_history.pop(0)

    def publish_twist(self, v_c=None):
        return twist_from_velocity(self.v_c if v_c is None else v_c, self.params.max_velocity)

    # -- rotation compensation: 4 (or any number of) candidate views against the one goal image
    ROTATION_SEARCH_PAIRS = 48                            # vitvs_v2.py:1158

    def best_rotation(self, candidate_frames: Sequence, generator: Optional[torch.Generator] = None):
        """Index of the candidate current frame whose selected correspondences are most similar on average, and the
        scores (reference: find_and_set_best_pose — ``num_pairs`` raised to 48 for these calls, score =
        ``sim_selected_12.mean()``, the first best wins, failed views are skipped), evaluated as ONE batch of
        ``len(candidate_frames)`` pairs sharing the goal forward.  Returns ``(None, scores)`` if every view failed."""
        eng = self.engine
        n = len(candidate_frames)
        if eng.max_pairs < n:
            raise ValueError("engine.max_pairs is smaller than the number of candidates")
        k = self.ROTATION_SEARCH_PAIRS
        if eng.max_rows < k:
            raise ValueError(f"engine.max_rows must be >= {k} for the rotation search")
        dev = eng.device
        frames = self._path_frames(self.goal_image, *candidate_frames)
        des = torch.as_tensor(frames[0]).to(dev)[None]
        cur = torch.stack([torch.as_tensor(f).to(dev) for f in frames[1:]])
        z = np.zeros((n, self.params.v_max, self.params.u_max), np.uint16)
        order = torch.stack([torch.randperm(eng.tokens, generator=generator) for _ in range(n)]).to(torch.int32)
        _, st = eng.compute_velocity(cur, des, z, self.params.intrinsics(), mode=_lib.SELECT_ORDER, selection=order,
                                     des_shared=True, num_pairs=k)
        det = eng.last_details(n)
        status = st.cpu().numpy()
        scores, best, best_mean = [], None, float("-inf")
        for b in range(n):
            m = int(det["info"][b, 3])
            if status[b] == _lib.STATUS_NO_CORRESPONDENCE or m == 0:
                scores.append(None)                       # "Feature detection failed for this rotation"
                continue
            mean = float(det["feat"][b, :m, 3].astype(np.float32).mean())
            scores.append(mean)
            if mean > best_mean:
                best_mean, best = mean, b
        return best, scores


# ------------------------------------------------------------------------------------------ several cameras, one GPU
class MultiController:
    """N cameras on ONE GPU (BASELINE.json configs[3], the 8-camera rig, when the rig has fewer GPUs than cameras).

    The reference runs one ``Controller`` per camera and process (vitvs_v2.py:702-819), each with its own goal image, EMA state,
    failure counter and history (:224, :325-343, :500-505).  Here every camera keeps exactly that state — ``self.cameras[i]`` IS a
    ``Controller`` and is fed through its own ``image_callback_*`` — while one round of updates for all of them goes to the GPU
    together:

      backend = ``Engine`` with ``max_pairs >= N``   ONE batched call per round: the N frame pairs run through the many-row
                                                     kernels as one launch chain (bench.py ``--pairs N``)
      backend = ``UpdatePipeline``                   one update per camera, ``depth`` of them in flight on separate handles and
                                                     streams (what bench.py measures ``value`` with); the cameras' inputs live in
                                                     device buffers of their own, so every (slot, camera) replays its captured graph

    Each camera's raw and smoothed ``v_c`` equal those of an independent ``Controller(Engine)`` fed the same frames and the same
    draw (tests/test_gpu_multi.py: bit for bit).  ``selection``: "order" (a fresh random visiting order per camera and round, drawn
    in camera order from ``generator`` / torch's global RNG, exactly the draws N ``Controller(selection="order")`` make when their
    ``ibvs()`` are called in camera order), "dense", or — per round, through ``ibvs(selection=[ids_0, ...])`` — explicit token ids.
    The reference's host-side sort + randperm draw ("reference") needs a host round trip per camera and stays with ``Controller``.
    """

    def __init__(self, backend, goal_images: Sequence, params: Optional[ServoParams] = None, selection: str = "order",
                 generator: Optional[torch.Generator] = None):
        from .pipeline import UpdatePipeline
        self.pipe = backend if isinstance(backend, UpdatePipeline) else None
        self.engines = list(backend.engines) if self.pipe is not None else [backend]
        self.engine = self.engines[0]
        if selection not in ("order", "dense"):
            raise ValueError('MultiController draws on the device: selection is "order" or "dense" (explicit ids per round: ibvs(selection=...))')
        self.selection, self.generator = selection, generator
        n = len(goal_images)
        if self.pipe is None and self.engine.max_pairs < n:
            raise ValueError(f"engine.max_pairs ({self.engine.max_pairs}) is smaller than the number of cameras ({n})")
        self.cameras = [Controller(self.engine, g, params, selection="order") for g in goal_images]
        self.params = self.cameras[0].params
        self._buffers = {}                                # pipeline mode: per-camera device inputs at stable addresses

    def __len__(self):
        return len(self.cameras)

    @property
    def v_c(self):
        """The cameras' smoothed twists (None until a camera's first successful update), like ``Controller.v_c``."""
        return [c.v_c for c in self.cameras]

    # -- inputs: camera i's ROS callbacks
    def image_callback_rgb(self, i: int, rgb_u8):
        self.cameras[i].image_callback_rgb(rgb_u8)

    def image_callback_depth(self, i: int, depth_u16):
        self.cameras[i].image_callback_depth(depth_u16)

    # -- one round
    def _arrays(self, imgs):
        out = []
        for img in imgs:
            if hasattr(img, "convert") and not isinstance(img, np.ndarray):   # PIL image
                img = np.asarray(img.convert("RGB"), dtype=np.uint8)
            out.append(img if torch.is_tensor(img) else np.asarray(img, dtype=np.uint8))
        return out

    def _frames_for(self, live):
        """(current frames, goal frames) of the live cameras as the engines take them: one geometry for all (the usual rig) goes in
        as it is and is resized inside the patch-row build (``Engine.set_frame_size``); anything else is resized per frame first."""
        cur = self._arrays([self.cameras[i].latest_pil_image for i in live])
        des = self._arrays([self.cameras[i].goal_image for i in live])
        shapes = {tuple(a.shape[:2]) for a in cur + des}
        ctl0 = self.cameras[0]
        if len(shapes) == 1 and next(iter(shapes)) not in ctl0._rejected_geometries:
            try:
                for e in self.engines:
                    e.set_frame_size(*next(iter(shapes)))
                return cur, des
            except VitvsError:
                ctl0._rejected_geometries.add(next(iter(shapes)))   # (see Controller._path_frames: never retried per step)
        for e in self.engines:
            e.set_frame_size()
        ctl = self.cameras[0]
        return [ctl._resized(a) for a in cur], [ctl._resized(a) for a in des]

    def ibvs(self, selection=None, want_features: bool = False):
        """One control-law step for every camera that has an image: each camera's ``v_c`` is updated (EMA) or left untouched on
        failure, exactly as its own ``Controller.ibvs()`` would.  Returns the per-camera ``detect_features`` results
        (``((s_uv_star, s_uv), sim)`` or ``(None, None)``; ``None`` for cameras without an image) when ``want_features``."""
        live = [i for i, c in enumerate(self.cameras) if c.latest_image is not None]
        results = [None] * len(self.cameras)
        if not live:
            return results if want_features else None
        p, eng = self.params, self.engine
        cur, des = self._frames_for(live)
        zeros = np.zeros((p.v_max, p.u_max), np.uint16)
        depth = [self.cameras[i].latest_image_depth if self.cameras[i].latest_image_depth is not None else zeros for i in live]
        k = self.cameras[0].num_pairs
        if selection is not None:
            mode, sel = _lib.SELECT_EXPLICIT, [np.asarray(selection[i]) for i in live]
        elif self.selection == "dense":
            mode, sel = _lib.SELECT_DENSE, None
        else:                                             # one fresh order per camera, drawn in camera order
            mode = _lib.SELECT_ORDER
            sel = [torch.randperm(eng.tokens, generator=self.generator).to(torch.int32) for _ in live]
        if self.pipe is None:
            stack = lambda xs: torch.stack([torch.as_tensor(x).to(eng.device) for x in xs])   # noqa: E731
            v, st = eng.compute_velocity(stack(cur), stack(des), np.stack(depth), p.intrinsics(), mode=mode,
                                         selection=(torch.stack(sel) if mode == _lib.SELECT_ORDER else sel), num_pairs=k)
            v, st = v.cpu().numpy(), st.cpu().numpy()
            det = eng.last_features(len(live)) if want_features else None
            failure = None
            for j, i in enumerate(live):
                # a camera's 10th consecutive failure raises (vitvs_v2.py:500-505) — after EVERY camera of the round has been
                # absorbed: N independent Controllers would each have taken their own step
                try:
                    ok = self.cameras[i]._absorb(v[j], int(st[j]))
                except RuntimeError as exc:
                    failure, ok = failure or exc, False
                if want_features:
                    results[i] = self.cameras[i]._features(det, j) if ok else (None, None)
                self.cameras[i]._law_step(ok)
            if failure is not None:
                raise failure
            return results if want_features else None
        # pipeline: `depth` cameras in flight at a time; a slot's outputs are read before the slot is used again
        pipe, dev = self.pipe, eng.device
        pending = []
        failures = []

        def collect(upto):
            while len(pending) > upto:
                i, t = pending.pop(0)
                v, st = pipe.result(t)
                try:
                    ok = self.cameras[i]._absorb(v.cpu().numpy()[0], int(st[0]))
                except RuntimeError as exc:               # raised once the round is complete and nothing is in flight (below)
                    failures.append(exc)
                    ok = False
                if want_features:
                    results[i] = self.cameras[i]._features(pipe.engines[t % pipe.depth].last_features(1), 0) if ok else (None, None)
                self.cameras[i]._law_step(ok)

        for j, i in enumerate(live):
            buf = self._buffers.setdefault(i, {})

            def stable(name, value, dtype=None):
                t_ = torch.as_tensor(value)
                t_ = t_ if dtype is None else t_.to(dtype)
                b_ = buf.get(name)
                if b_ is None or b_.shape != t_.shape or b_.dtype != t_.dtype:
                    b_ = torch.empty(t_.shape, dtype=t_.dtype, device=dev)
                    buf[name] = b_
                b_.copy_(t_, non_blocking=True)
                return b_
            c_ = stable("cur", cur[j])[None]
            d_ = stable("des", des[j])[None]
            z_ = stable("Z", depth[j])[None]
            k_ = stable("K", torch.tensor([p.intrinsics()], dtype=torch.float64))
            s_, n_ = None, None
            if mode == _lib.SELECT_ORDER:
                s_ = stable("order", sel[j])[None]
            elif mode == _lib.SELECT_EXPLICIT:
                ids = torch.zeros(k, dtype=torch.int32)
                got = torch.as_tensor(sel[j], dtype=torch.int32).flatten()[:k]
                ids[:got.numel()] = got
                s_ = stable("ids", ids)[None]
                n_ = stable("n_ids", torch.tensor([got.numel()], dtype=torch.int32))
            collect(pipe.depth - 1)
            pending.append((i, pipe.submit(c_, d_, z_, k_, mode, s_, n_, False, k)))
        collect(0)
        if failures:
            pipe.synchronize()                            # no slot still reads the cameras' input buffers when the caller sees it
            raise failures[0]
        return results if want_features else None

    def detect_features(self):
        """Every camera's ``Controller.detect_features()`` result for this round (and the law step, as ``ibvs`` does it)."""
        return self.ibvs(want_features=True)

    def publish_twist(self, i: int, v_c=None):
        return self.cameras[i].publish_twist(v_c)
