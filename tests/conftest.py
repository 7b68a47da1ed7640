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


@pytest.fixture(scope="session", autouse=True)
def _torch_first_on_the_gpu_box(request):
    """Some GPU tests place torch tensors next to the library's contexts (gathered images, halo arrays).  On this image torch's lazy
    CUDA initialisation fails ("No HIP GPUs are available") once another HIP context of the process has been created and destroyed,
    so on a GPU box torch initialises first — the order bench.py has anyway.  Nothing happens without a GPU or for CPU-only selections."""
    if "not gpu" in (request.config.getoption("-m") or ""):
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:       # torch missing or no device: the tests that need it will say so
        pass
