"""MI355X-native decode path for Dia-1.6B — host-side package.

``from dia_hip import Dia`` mirrors ``from dia.model import Dia`` of the reference
(dia/__init__.py:1-6).  Importing this package does not touch the GPU; the HIP library is
loaded on first use and its absence is an error, never a fallback.
"""

from .config import DiaConfig  # noqa: F401


def __getattr__(name):
    if name in ("Dia", "ComputeDtype"):
        from . import model as _m

        return getattr(_m, name)
    raise AttributeError(name)
