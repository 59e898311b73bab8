"""Import-path alias of the reference package: ``from dia.model import Dia`` resolves to the
MI355X-native implementation in ``dia_hip`` (reference dia/__init__.py:1-6 exports ``Dia``)."""
from dia_hip.model import Dia  # noqa: F401

__all__ = ["Dia"]
