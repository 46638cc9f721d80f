import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def native():
    """The HIP library behind the C ABI.  Building is part of the product; there is no fallback."""
    from htk_amd import build as nbuild, capi
    if not os.path.exists(capi.LIBPATH):
        nbuild.build()
    return capi
