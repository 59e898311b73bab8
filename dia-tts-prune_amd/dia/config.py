"""``dia.config`` alias (reference dia/config.py)."""
from dia_hip.config import DataConfig, DecoderConfig, DiaConfig, EncoderConfig, ModelConfig  # noqa: F401
