"""Isolated timing of the generic split-K weight-gradient kernel (cn_conv2d_wgrad) on the 1x1 layers of a DLA-34 step (batch 64).
    python tools/attic/wgrad1x1_bench.py [blocks ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from centernet_amd import _hip  # noqa: E402
from opbench import timeit  # noqa: E402

DEV = "cuda"
blocks = [int(a) for a in sys.argv[1:]] or [160, 1536]
dt = torch.bfloat16
code = _hip.dtype_code(dt)
for (N, HW, Ci, Co) in [(64, 128, 256, 2), (64, 128, 256, 80), (64, 128, 64, 64), (64, 64, 128, 128), (64, 128, 32, 64), (64, 32, 256, 256), (64, 64, 256, 128)]:
    ld = (Co + 15) // 16 * 16
    x = torch.randn(N, HW, HW, Ci, device=DEV).to(dt)
    dy = torch.randn(N, HW, HW, ld, device=DEV).to(dt)
    dwp = torch.zeros((Co + 31) // 32 * 32, Ci, device=DEV)
    for b in blocks:
        hk = _hip.Hooks().set(wgrad_blocks=b)
        us, mn = timeit(lambda: _hip.call("cn_conv2d_wgrad", x, dy, dwp, None, N, HW, HW, Ci, Ci, HW, HW, Co, ld, 1, 1, 1, 0, code, hooks=hk), n=10)
        nbytes = (x.numel() + dy.numel()) * 2
        print(f"wgrad 1x1 {Ci:3d}->{Co:3d} @{HW:3d}^2  blocks {b:5d}: {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s", flush=True)
    del x, dy, dwp
