import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dia-tts-prune_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture
def tuning():
    """set launch-heuristic overrides (dia_set_tuning) for one test; every knob touched is cleared afterwards"""
    from dia_hip import binding as hb
    touched = []

    def set_(name, value):
        touched.append(name)
        hb.set_tuning(name, value)

    yield set_
    for name in touched:
        hb.set_tuning(name, -1)
