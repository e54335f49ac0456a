"""Handle to the HIP hot path (libvitvs_hip.so) for one extractor configuration on one GPU.

PyTorch is used here only as plumbing: device buffers (``tensor.data_ptr()``), the current HIP
stream, and host-side weight preparation.  All arithmetic of the path runs in the HIP kernels
behind the C ABI (include/vitvs.h); nothing here falls back to torch ops or to the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .config import ServoParams, ViTConfig
from .weights import check_state_dict, resample_pos_embed


class VitvsError(RuntimeError):
    pass


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream_ptr(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Engine:
    """One ``vitvs_handle``: device weights + workspaces for up to ``max_pairs`` frame pairs."""

    def __init__(self, cfg: ViTConfig, params: ServoParams = None, *, precision: str = "fp32", max_pairs: int = 1,
                 max_rows: Optional[int] = None, binned: Optional[bool] = None, device=None):
        if not torch.cuda.is_available():
            raise VitvsError("no HIP device: the ViT-VS hot path has no CPU fallback")
        self.lib = _lib.load()
        self.cfg = cfg
        self.params = params or ServoParams(dino_input_size=cfg.img_size)
        # "f16x2": split-f16 (include/vitvs.h VITVS_F16X2) — fp32-class results on the f16 matrix cores, the parity mode at servo rate
        self.precision = {"fp32": _lib.F32, "f32": _lib.F32, "bf16": _lib.BF16, "fp16": _lib.F16, "f16": _lib.F16,
                          "f16x2": _lib.F16X2, "split-f16": _lib.F16X2}[precision]
        self.precision_name = {_lib.F32: "fp32", _lib.BF16: "bf16", _lib.F16: "fp16", _lib.F16X2: "f16x2"}[self.precision]
        self.binned = self.params.use_feature_binning if binned is None else bool(binned)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.max_pairs = int(max_pairs)
        # capacity in feature pairs per frame pair: the reference's rotation search raises num_pairs to 48 for its calls
        # (vitvs_v2.py:1151-1189), so that is the default floor
        self.max_rows = int(max_rows) if max_rows is not None else max(int(self.params.num_pairs), 48)
        c = _lib.VitvsConfig()
        c.abi_version = _lib.ABI_VERSION
        c.img_size, c.patch, c.stride, c.dim = cfg.img_size, cfg.patch, cfg.stride, cfg.dim
        c.heads, c.blocks, c.layerscale = cfg.heads, cfg.blocks_run, int(cfg.layerscale)
        for i in range(3):
            c.mean[i] = cfg.mean[i]
            c.std[i] = cfg.std[i]
        c.ln_eps = cfg.ln_eps
        c.precision = self.precision
        c.binned = int(self.binned)
        c.num_pairs = self.params.num_pairs
        c.u_max, c.v_max = self.params.u_max, self.params.v_max
        c.lambda_ = self.params.lambda_
        c.max_pairs, c.max_rows = self.max_pairs, self.max_rows
        self._c = c
        self.handle = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.vitvs_create(C.byref(c), C.byref(self.handle))
        if rc != 0:
            raise VitvsError(f"vitvs_create failed ({rc}): {_lib.last_error(None)}")
        self.frame_size = (cfg.img_size, cfg.img_size)   # geometry of the frames the calls take (set_frame_size)
        self._last_host_pairs = 1                         # pairs of the last host-pointer velocity call (reselect_host)
        self.tokens = self.lib.vitvs_tokens(self.handle)
        self.desc_dim = self.lib.vitvs_desc_dim(self.handle)
        assert self.tokens == cfg.tokens

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.vitvs_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc < 0:
            raise VitvsError(f"{what} failed ({rc}): {_lib.last_error(self.handle)}")
        return rc

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> "Engine":
        """Upload a DINO / timm / DINOv2 state dict (fp32).  ``pos_embed`` is resampled to this
        handle's token grid on the host first (reference: dinov2_extractor.py:94-118)."""
        check_state_dict(self.cfg, sd)
        for name, t in sd.items():
            if name == "pos_embed":
                t = resample_pos_embed(t, self.cfg.grid)
            elif name.startswith("blocks."):
                if int(name.split(".")[1]) >= self.cfg.blocks_run:
                    continue
            elif name not in ("patch_embed.proj.weight", "patch_embed.proj.bias", "cls_token"):
                continue
            a = np.ascontiguousarray(t.detach().to(torch.float32).cpu().numpy())
            rc = self.lib.vitvs_set_tensor(self.handle, name.encode(), a.ctypes.data_as(C.c_void_p), a.size)
            self._check(rc, f"vitvs_set_tensor({name})")
        self._check(self.lib.vitvs_weights_ready(self.handle), "vitvs_weights_ready")
        return self

    def share_weights(self, owner: "Engine") -> "Engine":
        """Borrow ``owner``'s device weights instead of uploading a copy (``vitvs_share_weights``): same network, input
        geometry and precision; ``owner`` must stay alive as long as this engine."""
        self._check(self.lib.vitvs_share_weights(self.handle, owner.handle), "vitvs_share_weights")
        self._weights_owner = owner                      # keeps the owner (and its device memory) alive
        return self

    # ------------------------------------------------------------------ helpers
    def _frames(self, frames) -> torch.Tensor:
        t = torch.as_tensor(frames)
        if t.dtype != torch.uint8:
            raise VitvsError("frames must be uint8 RGB, HWC")
        if t.dim() == 3:
            t = t.unsqueeze(0)
        fh, fw = self.frame_size
        if tuple(t.shape[1:]) != (fh, fw, 3):
            raise VitvsError(f"frames must be [n,{fh},{fw},3], got {tuple(t.shape)}")
        return t.to(self.device).contiguous()

    def set_frame_size(self, height: Optional[int] = None, width: Optional[int] = None) -> "Engine":
        """Declare the geometry of the frames handed to this engine from now on: camera frames uint8 [height, width, 3]
        instead of [S, S, 3].  The reference's ``image.resize((S, S))`` (PIL bicubic, vitvs_v2.py:474-475) then happens
        inside the launch that builds the patch rows (bit-identical to PIL, no resized image in memory).  No arguments:
        back to frames at S x S."""
        s = self.cfg.img_size
        h, w = (0, 0) if height is None or (height, width) == (s, s) else (int(height), int(width))
        rc = self.lib.vitvs_set_frame_size(self.handle, h, w)
        self._check(rc, "vitvs_set_frame_size")
        self.frame_size = (h, w) if h else (s, s)
        return self

    # ------------------------------------------------------------------ seams
    def forward_tokens(self, frames) -> torch.Tensor:
        """Residual stream after block ``layer``: float32 [n, 1+T, D] (the hooked tensor,
        reference: dinov2_extractor.py:198-199)."""
        f = self._frames(frames)
        n = f.shape[0]
        out = torch.empty((n, self.cfg.seq, self.cfg.dim), dtype=torch.float32, device=self.device)
        rc = self.lib.vitvs_forward_tokens_dev(self.handle, n, _ptr(f), _ptr(out), _stream_ptr(self.device))
        self._check(rc, "vitvs_forward_tokens_dev")
        return out

    def resize_frames(self, frames) -> torch.Tensor:
        """Camera frames uint8 [n,h,w,3] (or [h,w,3]) -> uint8 [n,S,S,3] on the device, bit-identical to
        ``PIL.Image.resize((S, S))`` (the reference's resize in front of the path, vitvs_v2.py:474-475)."""
        f = torch.as_tensor(np.ascontiguousarray(frames)) if not torch.is_tensor(frames) else frames
        if f.dim() == 3:
            f = f[None]
        if f.dtype != torch.uint8 or f.dim() != 4 or f.shape[-1] != 3:
            raise ValueError("frames must be uint8 [n,h,w,3]")
        f = f.to(self.device).contiguous()
        n, h, w, _ = f.shape
        s = self.cfg.img_size
        out = torch.empty((n, s, s, 3), dtype=torch.uint8, device=self.device)
        rc = self.lib.vitvs_resize_frames_dev(self.handle, n, _ptr(f), h, w, _ptr(out), _stream_ptr(self.device))
        self._check(rc, "vitvs_resize_frames_dev")
        return out

    FACETS = ("query", "key", "value", "token")

    def extract_descriptors(self, frames, facet: str = "token", bin: Optional[bool] = None,
                            include_cls: bool = False) -> torch.Tensor:
        """``ViTExtractor.extract_descriptors(batch, layer, facet, bin, include_cls)`` (dinov2_extractor.py:313-337):
        [n,1,T,D], [n,1,T,9D] with ``bin`` (3x3 log-bin of that facet), [n,1,1+T,D] with ``include_cls``.  facet 'token' (the
        servo path's choice) or 'query' / 'key' / 'value' (descriptor index d*H + h like the reference).  ``bin=None``: the
        engine's ``use_feature_binning`` for the token facet, False for the others.  ``bin`` with ``include_cls`` raises the
        reference's AssertionError; an unknown facet its TypeError-like message."""
        if facet not in self.FACETS:
            raise TypeError(f"{facet} is not a supported facet.")                # the reference's message
        binned = (self.binned if facet == "token" else False) if bin is None else bool(bin)
        if binned and include_cls:
            raise AssertionError("bin = True and include_cls = True are not supported together, set one of them False.")
        f = self._frames(frames)
        n = f.shape[0]
        if facet == "token" and not include_cls and binned == self.binned:      # the velocity path's own descriptors
            out = torch.empty((n, 1, self.tokens, self.desc_dim), dtype=torch.float32, device=self.device)
            rc = self.lib.vitvs_extract_descriptors_dev(self.handle, n, _ptr(f), _ptr(out), _stream_ptr(self.device))
            self._check(rc, "vitvs_extract_descriptors_dev")
            return out
        rows = self.tokens + (1 if include_cls else 0)
        out = torch.empty((n, 1, rows, self.cfg.dim * (9 if binned else 1)), dtype=torch.float32, device=self.device)
        rc = self.lib.vitvs_extract_descriptors_ex_dev(self.handle, n, _ptr(f), self.FACETS.index(facet), int(binned),
                                                       int(include_cls), _ptr(out), _stream_ptr(self.device))
        self._check(rc, "vitvs_extract_descriptors_ex_dev")
        return out

    def extract_saliency_maps(self, frames, head_idxs=(0, 2, 4, 5)) -> torch.Tensor:
        """``ViTExtractor.extract_saliency_maps(batch)`` (dinov2_extractor.py:339-353): the class token's attention over the
        patch tokens in block ``layer`` (the reference hooks block 11), averaged over ``head_idxs`` and min-max normalised per
        image: float32 [n, T] in [0, 1].  Like the reference, only for ``dino_vits8`` (its assertion, same message)."""
        assert self.cfg.model_type == "dino_vits8", "saliency maps are supported only for dino_vits model_type."
        f = self._frames(frames)
        n = f.shape[0]
        heads = (C.c_int32 * len(head_idxs))(*[int(i) for i in head_idxs])
        out = torch.empty((n, self.tokens), dtype=torch.float32, device=self.device)
        rc = self.lib.vitvs_extract_saliency_dev(self.handle, n, _ptr(f), len(head_idxs), heads, _ptr(out),
                                                 _stream_ptr(self.device))
        self._check(rc, "vitvs_extract_saliency_dev")
        return out

    def correspond(self, desc1: torch.Tensor, desc2: torch.Tensor, want_matrix: bool = False):
        """Similarity + argmax stage of find_correspondences_batch on [T,D] descriptors."""
        d1 = desc1.to(self.device, torch.float32).contiguous()
        d2 = desc2.to(self.device, torch.float32).contiguous()
        t, d = d1.shape
        pad = (-d) % 32
        if pad:  # zero columns leave cosine similarities unchanged
            d1 = torch.nn.functional.pad(d1, (0, pad))
            d2 = torch.nn.functional.pad(d2, (0, pad))
        nn1 = torch.empty(t, dtype=torch.int32, device=self.device)
        nn2 = torch.empty(t, dtype=torch.int32, device=self.device)
        sim1 = torch.empty(t, dtype=torch.float32, device=self.device)
        smat = torch.empty((t, t), dtype=torch.float32, device=self.device) if want_matrix else None
        rc = self.lib.vitvs_correspond_dev(self.handle, t, d + pad, _ptr(d1), _ptr(d2), _ptr(nn1), _ptr(nn2),
                                           _ptr(sim1), _ptr(smat), _stream_ptr(self.device))
        self._check(rc, "vitvs_correspond_dev")
        return (nn1, nn2, sim1, smat) if want_matrix else (nn1, nn2, sim1)

    def _num_pairs(self, num_pairs) -> int:
        k = int(self.params.num_pairs if num_pairs is None else num_pairs)
        if not 1 <= k <= self.max_rows:
            raise VitvsError(f"num_pairs {k} outside 1..max_rows ({self.max_rows})")
        return k

    def _selection_args(self, mode, selection, n_pairs, tokens, num_pairs=None):
        if mode == _lib.SELECT_DENSE:
            return None, None
        if selection is None:
            raise VitvsError("this selection mode needs a selection array")
        if mode == _lib.SELECT_EXPLICIT:
            k = self._num_pairs(num_pairs)
            sel = torch.full((n_pairs, k), 0, dtype=torch.int32)
            cnt = torch.zeros(n_pairs, dtype=torch.int32)
            rows = selection if isinstance(selection, (list, tuple)) else [selection]
            if len(rows) != n_pairs:
                raise VitvsError("one id list per pair expected")
            for b, ids in enumerate(rows):
                ids = torch.as_tensor(ids, dtype=torch.int32).flatten()[:k]
                sel[b, :ids.numel()] = ids
                cnt[b] = ids.numel()
            return sel.to(self.device), cnt.to(self.device)
        order = torch.as_tensor(selection, dtype=torch.int32).reshape(n_pairs, tokens)
        return order.to(self.device).contiguous(), None

    def servo_from_nn(self, nn_1, nn_2, sim_1, depth, K, mode=_lib.SELECT_DENSE, selection=None, num_pairs=None):
        """Control law on given nearest-neighbour tables (one pair); ``num_pairs`` as in ``compute_velocity``."""
        nn1 = torch.as_tensor(nn_1).to(self.device, torch.int32).contiguous()
        nn2 = torch.as_tensor(nn_2).to(self.device, torch.int32).contiguous()
        s1 = torch.as_tensor(sim_1).to(self.device, torch.float32).contiguous()
        t = nn1.numel()
        z = None if depth is None else torch.as_tensor(depth).to(self.device).contiguous()
        if z is not None and (z.dtype != torch.uint16 or tuple(z.shape[-2:]) != (self.params.v_max, self.params.u_max)):
            raise VitvsError("depth must be uint16 [v_max,u_max]")
        kk = torch.as_tensor(K, dtype=torch.float64).reshape(1, 4).to(self.device)
        k = self._num_pairs(num_pairs)
        sel, cnt = self._selection_args(mode, selection, 1, t, k)
        v = torch.zeros((1, 6), dtype=torch.float64, device=self.device)
        st = torch.zeros(1, dtype=torch.int32, device=self.device)
        n_sel = int(cnt[0].item()) if cnt is not None else 0
        rc = self.lib.vitvs_servo_from_nn_dev(self.handle, t, _ptr(nn1), _ptr(nn2), _ptr(s1), _ptr(z), _ptr(kk), mode,
                                              _ptr(sel), n_sel, k, _ptr(v), _ptr(st), _stream_ptr(self.device))
        self._check(rc, "vitvs_servo_from_nn_dev")
        self._last_tokens = t
        return v[0], st[0]

    # ------------------------------------------------------------------ the hot path
    def set_goal(self, I_des) -> "Engine":
        """Forward the goal frame(s) once and keep their descriptors in the handle (``vitvs_set_goal_dev``): later
        ``compute_velocity(..., I_des=None, ...)`` calls forward only the current frames.  One frame per pair of the later
        calls, or one frame for ``des_shared`` calls.  The reference recomputes the goal every update
        (vitvs_v2.py:482-487); the cache is for servo loops whose goal image does not change, and any call that forwards
        other frames through the engine drops it."""
        des = self._frames(I_des)
        self._check(self.lib.vitvs_set_goal_dev(self.handle, int(des.shape[0]), _ptr(des), _stream_ptr(self.device)), "vitvs_set_goal_dev")
        return self

    def compute_velocity_dev(self, I_cur: torch.Tensor, I_des: Optional[torch.Tensor], Z: Optional[torch.Tensor],
                             K: torch.Tensor, mode: int = _lib.SELECT_DENSE, selection: Optional[torch.Tensor] = None,
                             n_selected: Optional[torch.Tensor] = None, des_shared: bool = False,
                             out_v: Optional[torch.Tensor] = None, out_status: Optional[torch.Tensor] = None,
                             num_pairs: int = 0):
        """Device-resident call: every argument is a CUDA tensor already laid out as the C ABI
        wants it; work is enqueued on the current stream and nothing synchronises.  ``num_pairs`` = feature pairs of the
        law for this call (0: the engine's default)."""
        n = I_cur.shape[0]
        v = out_v if out_v is not None else torch.empty((n, 6), dtype=torch.float64, device=self.device)
        st = out_status if out_status is not None else torch.empty(n, dtype=torch.int32, device=self.device)
        rc = self.lib.vitvs_compute_velocity_dev(self.handle, n, _ptr(I_cur), _ptr(I_des), int(des_shared), _ptr(Z),
                                                 _ptr(K), mode, _ptr(selection), _ptr(n_selected), int(num_pairs), _ptr(v),
                                                 _ptr(st), _stream_ptr(self.device))
        self._check(rc, "vitvs_compute_velocity_dev")
        self._last_tokens = self.tokens
        return v, st

    def compute_velocity(self, I_cur, I_des, Z, K, mode: int = _lib.SELECT_DENSE, selection=None,
                         des_shared: bool = False, num_pairs: Optional[int] = None):
        """Convenience form: numpy / CPU inputs are moved to the device, then the device path runs.  ``num_pairs``:
        the reference's ``Controller.num_pairs`` for this call (default: the engine's parameters)."""
        cur = self._frames(I_cur)
        des = self._frames(I_des) if I_des is not None else None   # None: the goal cached by set_goal()
        n = cur.shape[0]
        if des is not None and des.shape[0] != (1 if des_shared else n):
            raise VitvsError("I_des must hold one frame per pair (or one frame when des_shared)")
        z = None
        if Z is not None:
            z = torch.as_tensor(Z)
            if z.dtype != torch.uint16:
                raise VitvsError("Z must be the sensor's uint16 millimetre image")
            z = z.reshape(n, self.params.v_max, self.params.u_max).to(self.device).contiguous()
        kk = torch.as_tensor(K, dtype=torch.float64).reshape(-1, 4)
        if kk.shape[0] == 1 and n > 1:
            kk = kk.expand(n, 4)
        kk = kk.contiguous().to(self.device)
        k = self._num_pairs(num_pairs)
        sel, cnt = self._selection_args(mode, selection, n, self.tokens, k)
        return self.compute_velocity_dev(cur, des, z, kk, mode, sel, cnt, des_shared, num_pairs=k)

    def compute_velocity_host(self, I_cur, I_des, Z, K, mode: int = _lib.SELECT_DENSE, selection=None, n_selected=None,
                              des_shared: bool = False, num_pairs: Optional[int] = None):
        """The host-pointer entry point (``vitvs_compute_velocity``): numpy arrays in — uint8 frames [n, H, W, 3] in the
        engine's current frame geometry, uint16 depth [n, v_max, u_max] (or None), intrinsics [n, 4] or [4], an int32
        selection laid out for ``mode`` — numpy ``(v_c [n, 6] float64, status [n] int32)`` out, ONE synchronous C call: what
        the reference's ``detect_features`` + ``ibvs`` do per update with the arrays its callbacks hold
        (vitvs_v2.py:464-523, 588-632).  No torch tensor is created on this path."""
        cur = np.ascontiguousarray(I_cur, dtype=np.uint8)
        if cur.ndim == 3:
            cur = cur[None]
        n = cur.shape[0]
        des = None
        if I_des is not None:
            des = np.ascontiguousarray(I_des, dtype=np.uint8)
            if des.ndim == 3:
                des = des[None]
            if des.shape[0] != (1 if des_shared else n) or des.shape[1:] != cur.shape[1:]:
                raise VitvsError("I_des must hold one frame per pair (or one frame when des_shared), in the geometry of I_cur")
        if tuple(cur.shape[1:3]) != tuple(self.frame_size) or cur.shape[3] != 3:
            raise VitvsError(f"frames are {tuple(cur.shape[1:])}, the engine expects {tuple(self.frame_size)} x 3 (set_frame_size)")
        z = None
        if Z is not None:
            z = np.ascontiguousarray(Z)
            if z.dtype != np.uint16 or z.size != n * self.params.v_max * self.params.u_max:
                raise VitvsError("Z must be the sensor's uint16 millimetre image(s) [v_max, u_max]")
        kk = np.ascontiguousarray(np.broadcast_to(np.asarray(K, np.float64).reshape(-1, 4), (n, 4)))
        k = self._num_pairs(num_pairs)
        sel = cnt = None
        if mode == _lib.SELECT_EXPLICIT:
            rows = selection if isinstance(selection, (list, tuple)) else [selection]
            if len(rows) != n:
                raise VitvsError("one id list per pair expected")
            sel = np.zeros((n, k), np.int32)
            cnt = np.zeros(n, np.int32)
            for b, ids in enumerate(rows):
                ids = np.asarray(ids, np.int32).reshape(-1)[:k]
                sel[b, :ids.size] = ids
                cnt[b] = ids.size
        elif mode == _lib.SELECT_ORDER:
            sel = np.ascontiguousarray(np.asarray(selection, np.int32).reshape(n, self.tokens))
        v = np.zeros((n, 6), np.float64)
        st = np.zeros(n, np.int32)
        p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731
        rc = self.lib.vitvs_compute_velocity(self.handle, n, p(cur), p(des), int(des_shared), p(z), p(kk), mode, p(sel), p(cnt), k,
                                             p(v), p(st))
        self._check(rc, "vitvs_compute_velocity")
        self._last_tokens = self.tokens
        self._last_host_pairs = n
        return v, st

    def reselect_host(self, mode: int, selection=None, num_pairs: Optional[int] = None):
        """``vitvs_reselect``: the control law again, for another selection, on what the last ``compute_velocity_host`` left in
        the handle (arg-max keys, depth image, intrinsics) — the second half of the reference's update when the draw happens on
        the host between the correspondence and the law (vitvs_v2.py:127-141).  numpy ``(v_c [n, 6], status [n])``."""
        n = self._last_host_pairs
        k = self._num_pairs(num_pairs)
        sel = cnt = None
        if mode == _lib.SELECT_EXPLICIT:
            rows = selection if isinstance(selection, (list, tuple)) else [selection]
            if len(rows) != n:
                raise VitvsError("one id list per pair expected")
            sel = np.zeros((n, k), np.int32)
            cnt = np.zeros(n, np.int32)
            for b, ids in enumerate(rows):
                ids = np.asarray(ids, np.int32).reshape(-1)[:k]
                sel[b, :ids.size] = ids
                cnt[b] = ids.size
        elif mode == _lib.SELECT_ORDER:
            sel = np.ascontiguousarray(np.asarray(selection, np.int32).reshape(n, self.tokens))
        v = np.zeros((n, 6), np.float64)
        st = np.zeros(n, np.int32)
        p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731
        self._check(self.lib.vitvs_reselect(self.handle, mode, p(sel), p(cnt), k, p(v), p(st)), "vitvs_reselect")
        return v, st

    def last_tables(self, n_pairs: int = 1) -> dict:
        """``nn_1``, ``nn_2`` (int32) and ``sim_1`` (float32) [n, T] of the last servo call; after ``compute_velocity_host`` they
        come from host memory (the handle's pinned block), no device call."""
        t = getattr(self, "_last_tokens", self.tokens)
        nn1 = np.empty((n_pairs, t), np.int32)
        nn2 = np.empty((n_pairs, t), np.int32)
        sim1 = np.empty((n_pairs, t), np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        rc = self.lib.vitvs_last_details(self.handle, n_pairs, p(nn1), p(nn2), p(sim1), None, None, None, None, None)
        self._check(rc, "vitvs_last_details")
        return dict(nn_1=nn1, nn_2=nn2, sim_1=sim1)

    def last_features(self, n_pairs: int = 1) -> dict:
        """``info``, ``s_uv`` and ``feat`` of the last servo call — what ``detect_features`` returns is made of
        (vitvs_v2.py:511-553) — without the arg-max tables and ``L_e`` that ``last_details`` also fetches.  After
        ``compute_velocity_host`` they are already in host memory (the handle's pinned block): no device call at all."""
        r = self.max_rows
        info = np.empty((n_pairs, 8), np.int32)
        suv = np.empty((n_pairs, r, 4), np.int32)
        feat = np.empty((n_pairs, r, 4), np.float64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        rc = self.lib.vitvs_last_details(self.handle, n_pairs, None, None, None, p(info), None, p(suv), p(feat), None)
        self._check(rc, "vitvs_last_details")
        return dict(info=info, s_uv=suv, feat=feat)

    # ------------------------------------------------------------------ options
    def set_option(self, name: str, value: int) -> "Engine":
        """Per-handle options of include/vitvs.h: ``graph_replay`` (0 / 1), ``in_flight`` (updates run beside this handle's)."""
        self._check(self.lib.vitvs_set_option(self.handle, name.encode(), int(value)), f"vitvs_set_option({name})")
        return self

    # ------------------------------------------------------------------ measurement hooks
    def timing_enable(self, on: bool = True):
        self._check(self.lib.vitvs_timing_enable(self.handle, int(on)), "vitvs_timing_enable")

    def timing_collect(self) -> dict:
        """{kernel class: (total ms between its HIP event pairs, launches)} since the last collect."""
        n = self.lib.vitvs_timing_classes()
        ms = (C.c_double * n)()
        cnt = (C.c_int32 * n)()
        self._check(self.lib.vitvs_timing_collect(self.handle, n, ms, cnt), "vitvs_timing_collect")
        return {self.lib.vitvs_timing_class_name(i).decode(): (float(ms[i]), int(cnt[i])) for i in range(n)}

    def last_details(self, n_pairs: int = 1) -> dict:
        """Host copies of what the last servo call left on the device (synchronises)."""
        t, r = getattr(self, "_last_tokens", self.tokens), self.max_rows
        nn1 = np.empty((n_pairs, t), np.int32)
        nn2 = np.empty((n_pairs, t), np.int32)
        sim1 = np.empty((n_pairs, t), np.float32)
        info = np.empty((n_pairs, 8), np.int32)
        sel = np.empty((n_pairs, r), np.int32)
        suv = np.empty((n_pairs, r, 4), np.int32)
        feat = np.empty((n_pairs, r, 4), np.float64)
        L = np.empty((n_pairs, 7, 2 * r), np.float64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        rc = self.lib.vitvs_last_details(self.handle, n_pairs, p(nn1), p(nn2), p(sim1), p(info), p(sel), p(suv),
                                         p(feat), p(L))
        self._check(rc, "vitvs_last_details")
        return dict(nn_1=nn1, nn_2=nn2, sim_1=sim1, info=info, selected=sel, s_uv=suv, feat=feat, L=L)
