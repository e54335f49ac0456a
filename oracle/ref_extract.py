"""ORACLE tooling (build container only): load individual functions of the reference
out of its source files WITHOUT importing the modules (their top-level imports need
rospy / cv2 / timm / torchvision, which are absent offline).

The function bodies are compiled from the reference files where they lie under
/root/reference; nothing is copied into this repository.  Used by
oracle/make_golden.py to generate tests/golden/*.npz and by
tests/test_oracle_vs_reference.py (skipped when /root/reference is absent, e.g. on
the GPU box).
"""
from __future__ import annotations

import ast
import importlib.util
import math
import os
import types

import numpy as np
import torch

REFERENCE_ROOT = os.environ.get("VITVS_REFERENCE_ROOT", "/root/reference")
VITVS_PY = os.path.join(REFERENCE_ROOT, "catkin_ws/ibvs/src/vitvs_v2.py")
EXTRACTOR_PY = os.path.join(REFERENCE_ROOT, "catkin_ws/ibvs/src/dinov2_extractor.py")
ATTENTION_PY = os.path.join(REFERENCE_ROOT, "dino_patch/attention.py")


def available() -> bool:
    return os.path.isfile(VITVS_PY) and os.path.isfile(EXTRACTOR_PY)


class _QuietLog:
    """Stands in for the rospy logger object only (loginfo/logwarn/logerr are print-like)."""

    def __getattr__(self, name):
        return lambda *a, **k: None


def _compile_defs(path: str, func_names=(), class_methods=None, namespace=None):
    with open(path, "r") as fh:
        tree = ast.parse(fh.read(), filename=path)
    ns = dict(namespace or {})
    body = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in func_names:
            body.append(node)
        elif isinstance(node, ast.ClassDef) and class_methods and node.name in class_methods:
            keep = [n for n in node.body if isinstance(n, ast.FunctionDef) and n.name in class_methods[node.name]]
            body.append(ast.ClassDef(name=node.name, bases=[], keywords=[], body=keep, decorator_list=[]))
    mod = ast.Module(body=body, type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, path, "exec"), ns)
    return ns


def load_correspondence_functions():
    """chunk_cosine_sim, _to_cartesian, find_correspondences_batch (vitvs_v2.py:49-155)."""
    ns = _compile_defs(VITVS_PY, func_names=("chunk_cosine_sim", "_to_cartesian", "find_correspondences_batch"),
                       namespace={"torch": torch, "np": np})
    return ns["chunk_cosine_sim"], ns["_to_cartesian"], ns["find_correspondences_batch"]


def load_controller_math(**params):
    """An object carrying the reference Controller's pure-math methods
    (vitvs_v2.py:325-343, 525-553, 566-586, 634-659) and the given parameters as attributes."""
    methods = ("calculate_uv", "transform_to_real_world", "get_depth", "calculate_interaction_matrix",
               "initialize_ema", "update_ema")
    ns = _compile_defs(VITVS_PY, class_methods={"Controller": methods},
                       namespace={"torch": torch, "np": np, "rospy": _QuietLog()})
    obj = ns["Controller"]()
    for k, v in params.items():
        setattr(obj, k, v)
    return obj


def load_log_bin():
    """ViTExtractor._log_bin (dinov2_extractor.py:265-311) bound to a bare object."""
    ns = _compile_defs(EXTRACTOR_PY, class_methods={"ViTExtractor": ("_log_bin",)},
                       namespace={"torch": torch, "np": np, "math": math})
    return ns["ViTExtractor"]


def load_pos_enc_interpolator(patch_size: int, stride: int):
    """ViTExtractor._fix_pos_enc(...) (dinov2_extractor.py:85-120) -> unbound interpolate function."""
    ns = _compile_defs(EXTRACTOR_PY, class_methods={"ViTExtractor": ("_fix_pos_enc",)},
                       namespace={"torch": torch, "np": np, "math": math, "nn": torch.nn,
                                  "Tuple": tuple})
    fn = ns["ViTExtractor"].__dict__["_fix_pos_enc"]
    fn = getattr(fn, "__func__", fn)
    return fn(patch_size, (stride, stride))


def load_attention_module():
    """dino_patch/attention.py as a standalone module (its xformers import is optional)."""
    spec = importlib.util.spec_from_file_location("_ref_dino_attention", ATTENTION_PY)
    mod = importlib.util.module_from_spec(spec)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        spec.loader.exec_module(mod)
    return mod
