import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def siftlib():
    """libsiftmi.so, built in-tree; GPU tests fail loudly if it is absent or finds no device."""
    from sift_pyocl_amd import _lib
    L = _lib.lib()
    assert L.siftmi_device_count() >= 1, "no HIP device visible: GPU tests need an MI355X"
    return L
