import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests are skipped (not failed) where no GPU is visible, e.g. in the build container."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


def seeded(seed, *shape):
    return np.random.RandomState(seed).randn(*shape)


def unit(x):
    return x / np.linalg.norm(x, axis=-1, keepdims=True)


def contrastive_inputs(d):
    """Regenerates the inputs of tests/golden/contrastive.npz (rule in its `meta`)."""
    q = unit(seeded(1000 + d, 8, d))
    p = unit(seeded(1001 + d, 48, d))
    p[::6] = unit(p[::6] + (2.0 / np.sqrt(d)) * q)
    return q, p
