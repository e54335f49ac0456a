"""Model and servo configuration for the ViT-VS hot path.

The model names are the ones the reference's ``ViTExtractor`` accepts
(reference: catkin_ws/ibvs/src/dinov2_extractor.py:28-29, 57-83); the servo
defaults are the reference's config.yaml values
(reference: catkin_ws/ibvs/config/config.yaml:1-17, 38).
"""
from __future__ import annotations

import dataclasses
import math
from dataclasses import dataclass

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
HALF_MEAN = (0.5, 0.5, 0.5)
HALF_STD = (0.5, 0.5, 0.5)


@dataclass(frozen=True)
class ViTConfig:
    """Geometry of one extractor model at one input size.

    ``layer`` is the block whose output is the descriptor (reference:
    vitvs_v2.py:484, dinov2_extractor.py:226-229); blocks after it are never run.
    """
    model_type: str
    img_size: int          # S: side of the (square) ViT input
    patch: int             # p
    stride: int            # patch-embed stride (== patch unless the extractor's stride hack is used)
    dim: int               # D
    depth: int             # blocks in the checkpoint
    heads: int             # H (head dim is always 64 for this family)
    layerscale: bool       # DINOv2 layout (ls1/ls2 gammas)
    native_grid: int       # side of the grid pos_embed is stored at
    layer: int = 11
    mlp_ratio: int = 4
    ln_eps: float = 1e-6

    @property
    def grid(self) -> int:
        # reference: dinov2_extractor.py:262 (1 + (H - p) // stride)
        return 1 + (self.img_size - self.patch) // self.stride

    @property
    def tokens(self) -> int:  # T, patch tokens
        return self.grid * self.grid

    @property
    def seq(self) -> int:  # N = T + cls
        return self.tokens + 1

    @property
    def head_dim(self) -> int:
        return self.dim // self.heads

    @property
    def hidden(self) -> int:
        return self.dim * self.mlp_ratio

    @property
    def blocks_run(self) -> int:
        return self.layer + 1

    @property
    def mean(self):
        # reference: dinov2_extractor.py:49-50 (substring test on the model name)
        return IMAGENET_MEAN if "dino" in self.model_type else HALF_MEAN

    @property
    def std(self):
        return IMAGENET_STD if "dino" in self.model_type else HALF_STD

    @property
    def patch_k(self) -> int:
        return 3 * self.patch * self.patch

    def flops_per_image(self) -> float:
        """Algorithmic FLOPs of one forward through block ``layer``."""
        n, d, h = self.seq, self.dim, self.hidden
        per_block = 2 * n * d * 3 * d + 2 * 2 * n * n * d + 2 * n * d * d + 2 * 2 * n * d * h
        return 2.0 * self.tokens * self.patch_k * d + self.blocks_run * per_block

    def flops_per_pair(self, binned: bool = False) -> float:
        gram = 2.0 * self.tokens * self.tokens * self.dim * (9 if binned else 1)
        return 2 * self.flops_per_image() + gram

    def weight_elems(self) -> int:
        d, h = self.dim, self.hidden
        per_block = 3 * d * d + d * d + 2 * d * h
        return self.patch_k * d + self.blocks_run * per_block


_FAMILY = {
    # name: (patch, dim, depth, heads, layerscale, native image side)
    "dino_vits16": (16, 384, 12, 6, False, 224),
    "dino_vits8": (8, 384, 12, 6, False, 224),
    "dino_vitb16": (16, 768, 12, 12, False, 224),
    "dino_vitb8": (8, 768, 12, 12, False, 224),
    "vit_small_patch16_224": (16, 384, 12, 6, False, 224),
    "vit_small_patch8_224": (8, 384, 12, 6, False, 224),
    "vit_base_patch16_224": (16, 768, 12, 12, False, 224),
    "vit_base_patch8_224": (8, 768, 12, 12, False, 224),
    "dinov2_vits14": (14, 384, 12, 6, True, 518),
    "dinov2_vitb14": (14, 768, 12, 12, True, 518),
    "dinov2_vitl14": (14, 1024, 24, 16, True, 518),
}


def vit_config(model_type: str, img_size: int, stride: int | None = None, layer: int = 11) -> ViTConfig:
    if model_type not in _FAMILY:
        raise ValueError(f"unknown model_type {model_type!r}; known: {sorted(_FAMILY)}")
    patch, dim, depth, heads, ls, native = _FAMILY[model_type]
    stride = patch if stride is None else int(stride)
    if (patch // stride) * stride != patch:
        # reference: dinov2_extractor.py:137-138
        raise ValueError(f"stride {stride} should divide patch_size {patch}")
    if not 0 <= layer < depth:
        raise ValueError(f"layer {layer} outside 0..{depth - 1}")
    if img_size < patch:
        raise ValueError("image smaller than one patch")
    if (img_size - patch) % stride != 0:
        # the reference floors (dinov2_extractor.py:262, 1 + (H - p) // stride) and silently ignores the right / bottom
        # remainder; the device path needs the covered extent to be exact, so ask for the size that is actually used
        used = patch + ((img_size - patch) // stride) * stride
        raise ValueError(f"img_size {img_size} leaves a remainder for patch {patch} / stride {stride}: "
                         f"resize or crop to {used} (the extent the reference's floor would use)")
    return ViTConfig(model_type=model_type, img_size=int(img_size), patch=patch, stride=stride, dim=dim,
                     depth=depth, heads=heads, layerscale=ls, native_grid=native // patch, layer=layer)


# The five BASELINE.json configs + the reference's shipped default, by short key.
BASELINE_CONFIGS = {
    "vits16_224": ("dino_vits16", 224),       # configs[0]
    "vitb16_224": ("vit_base_patch16_224", 224),  # configs[1] and [3]
    "vitb8_448": ("dino_vitb8", 448),         # configs[2]
    "vitl14_518": ("dinov2_vitl14", 518),     # configs[4]
    "vits14_308": ("dinov2_vits14", 308),     # reference default (vitvs_v2.py:250, config.yaml:14)
}


def baseline_config(key: str) -> ViTConfig:
    name, size = BASELINE_CONFIGS[key]
    return vit_config(name, size)


@dataclass
class ServoParams:
    """Camera + control-law parameters (reference: config.yaml:1-17,38; vitvs_v2.py:278-283)."""
    u_max: int = 640
    v_max: int = 480
    f_x: float = 502.3016357421875
    f_y: float = 502.3016357421875
    lambda_: float = 0.03
    num_pairs: int = 24
    dino_input_size: int = 308
    use_feature_binning: bool = True
    ema_alpha: float = 0.8
    max_velocity: float = 1.0

    @property
    def c_x(self) -> float:
        return self.u_max / 2

    @property
    def c_y(self) -> float:
        return self.v_max / 2

    def intrinsics(self):
        return (self.f_x, self.f_y, self.c_x, self.c_y)

    def replace(self, **kw) -> "ServoParams":
        return dataclasses.replace(self, **kw)


def grid_side(tokens: int) -> int:
    # reference: vitvs_v2.py:75 (int(np.sqrt(T)); square grids only)
    return int(math.sqrt(tokens))


# ------------------------------------------------------------------------------------------------------------------
# The reference's configuration file (catkin_ws/ibvs/config/config.yaml, read by Controller.load_parameters,
# vitvs_v2.py:272-323): the data format on the caller's side of the path.
_REQUIRED_KEYS = (   # load_parameters indexes these with config[...]: a missing one is a KeyError there and here
    "u_max", "v_max", "f_x", "f_y", "lambda_", "min_error", "max_error", "num_pairs", "thresh_filter_keypoints",
    "dino_input_size", "use_feature_binning", "num_samples", "num_circles", "circle_radius_aug",
    "velocity_convergence_threshold", "velocity_threshold_translation", "velocity_threshold_rotation",
    "error_threshold_ratio", "error_threshold_absolute_translation", "error_threshold_absolute_rotation",
    "min_iterations", "max_iterations", "image_path")


@dataclass
class ReferenceConfig:
    """What a ``config.yaml`` of the reference says, split by who reads it here."""
    servo: ServoParams                      # the path: camera model, gain, pairs, input size, binning, EMA, clip
    max_iterations: int                     # ServoLoop(max_iterations=...)                       (vitvs_v2.py:412)
    max_velocity_vector_history: int        # Controller.max_velocity_vector_history              (vitvs_v2.py:627)
    image_path: str                         # goal image, relative to the reference's script directory (:323)
    extras: dict                            # loaded by the reference, never read on the path (sampling, thresholds ...)

    def apply(self, controller=None, loop=None):
        """Copy the run-loop settings onto a ``servo.Controller`` / ``loop.ServoLoop`` built from ``self.servo``."""
        if controller is not None:
            controller.max_velocity_vector_history = self.max_velocity_vector_history
        if loop is not None:
            loop.max_iterations = self.max_iterations
        return self


def load_reference_config(source) -> ReferenceConfig:
    """``source``: path of a YAML file in the reference's schema, or the mapping ``yaml.safe_load`` returned.

    Same required keys as ``Controller.load_parameters`` (a missing one raises ``KeyError`` naming it) and the same
    defaults for the optional ones: ``max_velocity`` 1.0, ``ema_alpha`` 0.1 (NOT the 0.8 the shipped file sets),
    ``max_velocity_vector_history`` 200, ``background_thresh`` 0.5 (vitvs_v2.py:287, 296, 316, 319)."""
    if isinstance(source, dict):
        cfg = dict(source)
    else:
        import yaml
        with open(source, "r") as fh:
            cfg = yaml.safe_load(fh)
        if not isinstance(cfg, dict):
            raise ValueError(f"{source}: not a mapping")
    for key in _REQUIRED_KEYS:
        if key not in cfg:
            raise KeyError(key)
    servo = ServoParams(u_max=int(cfg["u_max"]), v_max=int(cfg["v_max"]), f_x=float(cfg["f_x"]), f_y=float(cfg["f_y"]),
                        lambda_=float(cfg["lambda_"]), num_pairs=int(cfg["num_pairs"]),
                        dino_input_size=int(cfg["dino_input_size"]), use_feature_binning=bool(cfg["use_feature_binning"]),
                        ema_alpha=float(cfg.get("ema_alpha", 0.1)), max_velocity=float(cfg.get("max_velocity", 1.0)))
    used = {"u_max", "v_max", "f_x", "f_y", "lambda_", "num_pairs", "dino_input_size", "use_feature_binning", "ema_alpha",
            "max_velocity", "max_iterations", "max_velocity_vector_history", "image_path"}
    extras = {k: v for k, v in cfg.items() if k not in used}
    extras.setdefault("background_thresh", 0.5)
    return ReferenceConfig(servo=servo, max_iterations=int(cfg["max_iterations"]),
                           max_velocity_vector_history=int(cfg.get("max_velocity_vector_history", 200)),
                           image_path=str(cfg["image_path"]), extras=extras)
