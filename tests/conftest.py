import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def luts():
    """Stand-in Bruneton LUTs (Scene input shared by the oracle and the HIP path)."""
    from hobbyrenderer_amd import native
    return native.precompute_atmosphere()


@pytest.fixture(scope="session")
def gpu_available():
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        return False


def rel_l2_per_pixel(a, b):
    """Per-pixel relative L2 of RGB: |a-b|_2 / max(|b|_2, tiny)."""
    a = np.asarray(a, np.float64)[..., :3]
    b = np.asarray(b, np.float64)[..., :3]
    num = np.sqrt(((a - b) ** 2).sum(-1))
    den = np.maximum(np.sqrt((b ** 2).sum(-1)), 1e-12)
    return num / den
