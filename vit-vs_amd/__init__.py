"""vitvs_amd: MI355X-native ViT-feature visual-servoing hot path."""
import os as _os

# Kernel arguments in device memory: a launch then needs no PCIe read before its first wave starts.  With ~90
# dependent launches per servo update this is worth 18 % (measured); the HIP runtime reads the variable when it
# initialises, i.e. at the first HIP call of the process, so it must be set before that.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from .config import ViTConfig, ServoParams, vit_config, baseline_config, BASELINE_CONFIGS  # noqa: F401

__version__ = "0.1.0"
