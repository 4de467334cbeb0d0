import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


SMALL_K = (200.0, 200.0, 80.0, 60.0)
SMALL_SIZE = (120, 160)


@pytest.fixture(scope="session")
def small_window():
    """4 frames x 300 points, 5x5 patches, 120x160 images: the oracle finishes in milliseconds."""
    from photobundle_amd import synthetic
    return synthetic.make_window(n_frames=4, n_points=300, radius=2, size=SMALL_SIZE, K=SMALL_K)


@pytest.fixture(scope="session")
def small_window_huber():
    from photobundle_amd import synthetic
    return synthetic.make_window(n_frames=4, n_points=200, radius=1, size=SMALL_SIZE, K=SMALL_K, huber=0.05,
                                 visibility="causal", seed_offset=3)
