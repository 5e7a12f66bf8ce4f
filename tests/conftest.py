import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """The C-ABI library, built in-tree if it is not there yet (hipcc cross-compiles on CPU)."""
    from eqxvision_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from eqxvision_amd.build import build
        build()
    return _lib.load()
