import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def small_window():
    from dsopp_amd import synthetic as syn
    return syn.make_window(num_frames=4, num_points=240, width=320, height=240, seed=3)


@pytest.fixture(scope="session")
def tiny_window():
    from dsopp_amd import synthetic as syn
    return syn.make_window(num_frames=3, num_points=60, width=160, height=120, seed=5)


def pytest_sessionstart(session):
    """PyTorch carries its own HIP runtime.  When it initialises AFTER another HIP user of the process (libdsopp_hip.so) it
    reports "No HIP GPUs are available"; the other way round both work (bench.py imports torch first for the same reason).
    Tests that hand torch-allocated device buffers to the library therefore need torch's runtime up before the first call."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
