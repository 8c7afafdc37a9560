import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def strided(t, n=4096):
    f = t.detach().flatten()
    step = max(1, f.numel() // n)
    return f[::step][:n].clone()


def summary(t):
    d = t.detach().double().cpu()
    return np.array([d.sum().item(), d.abs().sum().item(), (d * d).sum().item()])
