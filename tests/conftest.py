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


def dcn_border_vectors():
    """Hand-computed known answers for DCNv2's zero-padded bilinear sampling at the image borders (value 0 if
    py <= -1 or py >= H or px <= -1 or px >= W; otherwise each of the four corners contributes only if it lies inside the image).
    Image: one channel, 4x4, x[h][w] = 4h + w + 1 (1..16).  One sample point per output pixel, taken by the CENTRE tap
    (k = 4, weight 1, mask 1; every other tap has weight 0), so output[h][w] = bilinear(x, py, px).
    -> (x [1,1,4,4], offset [1,18,4,4], mask [1,9,4,4], weight [1,1,3,3], expected [4,4]) as float64 tensors."""
    import torch
    x = (torch.arange(16, dtype=torch.float64) + 1).view(1, 1, 4, 4)
    cases = [  # (py, px, hand-computed value)
        (-0.5, 1.0, 0.5 * 2),                      # top edge: row -1 is outside, half of x[0][1]
        (3.5, 2.0, 0.5 * 15),                      # bottom edge: row 4 is outside, half of x[3][2]
        (1.0, -0.25, 0.75 * 5),                    # left edge: column -1 outside, 0.75 of x[1][0]
        (2.0, 3.75, 0.25 * 12),                    # right edge: column 4 outside, 0.25 of x[2][3]
        (-1.0, 2.0, 0.0),                          # py == -1 exactly: outside
        (4.0, 1.0, 0.0),                           # py == H exactly: outside
        (1.0, -1.0, 0.0),                          # px == -1 exactly
        (1.0, 4.0, 0.0),                           # px == W exactly
        (-0.5, -0.5, 0.25 * 1),                    # top-left corner: only x[0][0], weight 1/4
        (3.5, 3.5, 0.25 * 16),                     # bottom-right corner: only x[3][3]
        (1.25, 2.5, 0.375 * 7 + 0.375 * 8 + 0.125 * 11 + 0.125 * 12),   # interior, all four corners (= 8.5)
        (-0.75, 3.25, 0.25 * 0.75 * 4),            # top-right: rows -1|0 (lh .25), cols 3|4 -> only x[0][3] * lh * (1-lw)
        (3.25, -0.75, 0.75 * 0.25 * 13),           # bottom-left: rows 3|4 (1-lh .75), cols -1|0 (lw .25) -> x[3][0]
        (2.0, 1.0, 10.0),                          # integer interior point: x[2][1]
        (-1.5, 1.5, 0.0),                          # beyond the top
        (2.5, 5.0, 0.0),                           # beyond the right
    ]
    off = torch.zeros(1, 18, 4, 4, dtype=torch.float64)
    want = torch.zeros(4, 4, dtype=torch.float64)
    for n, (py, px, v) in enumerate(cases):
        h, w = divmod(n, 4)
        off[0, 8, h, w] = py - h        # centre tap (i = j = 1): py = h - 1 + 1 + dy
        off[0, 9, h, w] = px - w
        want[h, w] = v
    mask = torch.ones(1, 9, 4, 4, dtype=torch.float64)
    weight = torch.zeros(1, 1, 3, 3, dtype=torch.float64)
    weight[0, 0, 1, 1] = 1.0
    return x, off, mask, weight, want
