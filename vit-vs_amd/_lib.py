"""ctypes binding of libvitvs_hip.so (include/vitvs.h, include/vitvs_ops.h).

The library is built in-tree by ``vit-vs_amd/csrc/Makefile`` (``__graft_entry__.build()``).
There is no CPU fallback: if the shared object is missing, or it exports fewer symbols than the
headers declare, importing fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# VITVS_LIB points at another build of the same sources (A/B experiments); the default is the in-tree library
LIB_PATH = os.environ.get("VITVS_LIB") or os.path.join(_HERE, "libvitvs_hip.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

ABI_VERSION = 2
F32, BF16, F16, F16X2 = 0, 1, 2, 3
STATUS_OK, STATUS_NO_CORRESPONDENCE, STATUS_TOO_FEW, STATUS_NO_DEPTH = 0, 1, 2, 3
SELECT_EXPLICIT, SELECT_ORDER, SELECT_DENSE = 0, 1, 2


class VitvsConfig(C.Structure):
    """Mirror of ``struct vitvs_config`` (include/vitvs.h)."""
    _fields_ = [
        ("abi_version", C.c_int32),
        ("img_size", C.c_int32), ("patch", C.c_int32), ("stride", C.c_int32), ("dim", C.c_int32),
        ("heads", C.c_int32), ("blocks", C.c_int32), ("layerscale", C.c_int32),
        ("mean", C.c_float * 3), ("std", C.c_float * 3), ("ln_eps", C.c_float),
        ("precision", C.c_int32), ("binned", C.c_int32),
        ("num_pairs", C.c_int32), ("u_max", C.c_int32), ("v_max", C.c_int32),
        ("lambda_", C.c_double),
        ("max_pairs", C.c_int32), ("max_rows", C.c_int32),
    ]


_P = C.c_void_p
_I = C.c_int32
# name -> (restype, argtypes); every function declared in include/vitvs.h and include/vitvs_ops.h
PROTOTYPES = {
    "vitvs_abi_version": (_I, []),
    "vitvs_create": (_I, [C.POINTER(VitvsConfig), C.POINTER(_P)]),
    "vitvs_destroy": (None, [_P]),
    "vitvs_last_error": (C.c_char_p, [_P]),
    "vitvs_set_tensor": (_I, [_P, C.c_char_p, _P, C.c_int64]),
    "vitvs_weights_ready": (_I, [_P]),
    "vitvs_compute_velocity_dev": (_I, [_P, _I, _P, _P, _I, _P, _P, _I, _P, _P, _I, _P, _P, _P]),
    "vitvs_compute_velocity": (_I, [_P, _I, _P, _P, _I, _P, _P, _I, _P, _P, _I, _P, _P]),
    "vitvs_reselect": (_I, [_P, _I, _P, _P, _I, _P, _P]),
    "vitvs_set_goal_dev": (_I, [_P, _I, _P, _P]),
    "vitvs_set_goal": (_I, [_P, _I, _P]),
    "vitvs_extract_descriptors_dev": (_I, [_P, _I, _P, _P, _P]),
    "vitvs_forward_tokens_dev": (_I, [_P, _I, _P, _P, _P]),
    "vitvs_correspond_dev": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "vitvs_servo_from_nn_dev": (_I, [_P, _I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _P, _P, _P]),
    "vitvs_last_details": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "vitvs_set_option": (_I, [_P, C.c_char_p, C.c_int64]),
    "vitvs_share_weights": (_I, [_P, _P]),
    "vitvs_timing_enable": (_I, [_P, _I]),
    "vitvs_timing_classes": (_I, []),
    "vitvs_timing_class_name": (C.c_char_p, [_I]),
    "vitvs_timing_collect": (_I, [_P, _I, _P, _P]),
    "vitvs_tokens": (_I, [_P]),
    "vitvs_desc_dim": (_I, [_P]),
    "vitvs_extract_facet_dev": (_I, [_P, _I, _P, _I, _P, _P]),
    "vitvs_extract_descriptors_ex_dev": (_I, [_P, _I, _P, _I, _I, _I, _P, _P]),
    "vitvs_resize_frames_dev": (_I, [_P, _I, _P, _I, _I, _P, _P]),
    "vitvs_set_frame_size": (_I, [_P, _I, _I]),
    "vitvs_extract_saliency_dev": (_I, [_P, _I, _P, _I, _P, _P, _P]),
    "vitvs_op_linear": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "vitvs_op_linear_variant": (_I, [_I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "vitvs_op_linear_residual": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "vitvs_op_layernorm": (_I, [_I, _P, _P, _P, _P, _I, _I, C.c_float, _P]),
    "vitvs_op_attention": (_I, [_I, _P, _P, _I, _I, _I, _P]),
    "vitvs_op_attention_q": (_I, [_I, _P, _P, _I, _I, _I, _I, _P]),
    "vitvs_op_splitk_slices": (_I, [_I, _I, _I, _I]),
    "vitvs_op_plan_in_flight": (_I, [_I]),
    "vitvs_op_weight_exponent": (_I, [_I]),
    "vitvs_op_touch": (_I, [_P, C.c_int64, _I, _P]),
    "vitvs_op_linear_tile": (_I, [_I, _I, _I, _I, _I, _P]),
    "vitvs_op_linear_partial": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "vitvs_op_residual_ln": (_I, [_I, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, C.c_float, _P]),
}


class VitvsLibraryError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into libvitvs_hip.so (hipcc cross-compiles without a GPU)."""
    res = subprocess.run(["make", "-C", CSRC_DIR, "-j", str(min(8, os.cpu_count() or 1))],
                         capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-8000:])
    if res.returncode != 0 or not os.path.isfile(LIB_PATH):
        raise VitvsLibraryError("building libvitvs_hip.so failed (see output above)")
    return LIB_PATH


_lib = None


def load():
    """Load the shared object and bind every declared prototype.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise VitvsLibraryError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C vit-vs_amd/csrc`).  There is no CPU fallback for the hot path.")
    # torch first: its wheel bundles its own libamdhip64, and a process must end up with ONE HIP runtime.  Loaded after
    # torch, this library's DT_NEEDED libamdhip64 resolves to the copy torch already mapped; loaded before it, the system
    # copy is mapped, torch then brings its own, and the two runtimes do not see each other's devices and streams
    # (symptom: vitvs_create "no HIP device available" in a process that called build() — which loads the library —
    # before anything imported torch).  Python callers of this binding always use torch for device memory and streams.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise VitvsLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.vitvs_abi_version() != ABI_VERSION:
        raise VitvsLibraryError("libvitvs_hip.so ABI version does not match the Python binding")
    _lib = lib
    return lib


def last_error(handle=None) -> str:
    msg = load().vitvs_last_error(handle)
    return msg.decode("utf-8", "replace") if msg else ""
