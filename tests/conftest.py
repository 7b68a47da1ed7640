import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lut():
    """A small BRDF LUT from the oracle (shared by the oracle and the HIP path so shading parity is LUT independent)."""
    from oracle import oracle_lib
    return oracle_lib.brdf_lut(64, 64)
