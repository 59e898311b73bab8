"""``dia.model`` alias (reference dia/model.py): same public names, HIP-backed."""
from dia_hip.model import DEFAULT_SAMPLE_RATE, ComputeDtype, Dia  # noqa: F401
