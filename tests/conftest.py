import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "launcher: starts child ranks; scheduled before the tests that initialise HIP")


def pytest_collection_modifyitems(config, items):
    """Tests that launch child ranks run first, while the pytest process itself has not touched the GPU."""
    items.sort(key=lambda it: 0 if it.get_closest_marker("launcher") else 1)


@pytest.fixture(scope="session")
def oracle_c():
    import oracle
    return oracle.c()


@pytest.fixture(scope="session")
def oracle_ref():
    import oracle
    r = oracle.ref()
    if not r.available:
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    return r


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from rs_detection_amd import _lib
    _lib.load()  # fail loudly if the HIP extension is missing
    return torch.device("cuda:0")


def dota_boxes(rng, n, span=1024.0, wmin=10, wmax=160, hmax=64):
    """SURVEY 8(d) synthetic gt geometry: w>=h, theta in [-pi/4, 3pi/4)."""
    w = rng.uniform(wmin, wmax, n)
    h = rng.uniform(5, np.minimum(w, hmax))
    return np.stack([rng.uniform(0, span, n), rng.uniform(0, span, n), w, h,
                     rng.uniform(-np.pi / 4, 3 * np.pi / 4, n)], 1).astype(np.float32)


def s2anet_anchors():
    import oracle
    return np.concatenate([oracle.np_s2anet_grid_anchors((1024 // s, 1024 // s), s) for s in (8, 16, 32, 64, 128)])


def degenerate_boxes():
    """identical / nested / touching / 45-degree / zero-area / far apart."""
    return np.array([
        [10, 10, 20, 10, 0], [10, 10, 20, 10, 0], [10, 10, 10, 5, 0], [30, 10, 20, 10, 0],
        [10, 10, 20, 10, np.pi / 4], [10, 10, 20, 10, np.pi / 2], [10, 10, 1e-8, 1e-8, 0],
        [20, 10, 20, 10, 0], [10, 10, 20, 10, np.pi], [500, 500, 30, 30, 0.3], [10, 20, 20, 10, 0],
        [10, 10, 20, 10, 1e-4], [10.00001, 10, 20, 10, 0], [0, 0, 1, 1, 0], [.5, .5, 1, 2, 0],
    ], np.float32)
