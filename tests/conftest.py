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
