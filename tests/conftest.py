import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle_lib_built():
    import oracle_lib
    oracle_lib.build()
    return oracle_lib


@pytest.fixture(scope="session")
def gpu_available():
    if not _has_gpu():
        pytest.fail("GPU test selected but no HIP device is visible: the HIP path has no CPU fallback")
    return True


@pytest.fixture(scope="session")
def png_pair():
    import numpy as np
    from PIL import Image
    g = os.path.join(ROOT, "tests", "golden")
    f = lambda n: np.array(Image.open(os.path.join(g, n + ".png")))
    return [(f("1c"), f("1d")), (f("2c"), f("2d"))]
