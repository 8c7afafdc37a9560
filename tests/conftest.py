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


def assert_det_rank_tolerant(got, ref):
    """ctdet_decode outputs [B,K,6] whose top-K scores sit closer together than the 1e-4 heat-map tolerance: neighbouring
    ranks may swap.  Every reference detection must be present (same class, same box, same score); ranks may differ
    only between scores within 2e-4 (relative) of each other, and such near-ties may fall off the end of the top-K."""
    import numpy as np
    for b in range(ref.shape[0]):
        for i, r in enumerate(ref[b]):
            d = np.abs(got[b][:, :4] - r[:4]).max(1) + 1e3 * (got[b][:, 5] != r[5])
            j = int(d.argmin())
            cutoff = abs(r[4] - ref[b][-1, 4]) < 2e-4 * r[4]
            assert cutoff or (d[j] < 2e-3 + 1e-3 * np.abs(r[:4]).max() and abs(got[b][j, 4] - r[4]) < 1e-4), (b, i)
            assert cutoff or j == i or abs(ref[b][j, 4] - r[4]) < 2e-4 * r[4], (b, i, j)
