"""vitvs_amd: MI355X-native ViT-feature visual-servoing hot path."""
from .config import ViTConfig, ServoParams, vit_config, baseline_config, BASELINE_CONFIGS  # noqa: F401

__version__ = "0.1.0"
