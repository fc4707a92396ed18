import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session")
def gpu_required():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no GPU is visible")
    return True


@pytest.fixture(autouse=True)
def _gpu_tests_see_the_device_first(request):
    """Every `gpu` test goes through `gpu_required` before it touches the library: torch's bundled HIP runtime must come up before
    libcimpc_hip.so brings up the system one (the other order left `torch.cuda.is_available()` False for the rest of the process -
    seen when tests/test_gpu_round6.py, which does not name the fixture, ran first)."""
    if request.node.get_closest_marker("gpu") is not None:
        request.getfixturevalue("gpu_required")
