"""Isolated timing of cn_conv2d_wgrad on the 3x3 / stride-1 layers of a DLA-34 step (batch 64) at several split-K grid sizes
(cn_hooks.wgrad_blocks: 384 = what the train step uses next to the data-gradient chain, 1536 = alone).
    python tools/wgrad_bench.py [blocks ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from centernet_amd import _hip  # noqa: E402
from opbench import timeit  # noqa: E402

DEV = "cuda"
SHAPES = [(64, 128, 64, 64), (64, 128, 64, 256), (64, 128, 64, 27), (64, 64, 128, 128), (64, 32, 256, 256), (64, 16, 512, 512), (64, 64, 64, 27)]
blocks = [int(a) for a in sys.argv[1:]] or [384, 1536]
dt = torch.bfloat16
code = _hip.dtype_code(dt)
for (N, HW, Ci, Co) in SHAPES:
    ld = (Co + 15) // 16 * 16
    x = torch.randn(N, HW, HW, Ci, device=DEV).to(dt)
    dy = torch.randn(N, HW, HW, ld, device=DEV).to(dt)
    dwp = torch.zeros((Co + 31) // 32 * 32, 9 * Ci, device=DEV)
    flops = 2.0 * N * HW * HW * Ci * Co * 9
    for b in blocks:
        hk = _hip.Hooks().set(wgrad_blocks=b)
        us, mn = timeit(lambda: _hip.call("cn_conv2d_wgrad", x, dy, dwp, None, N, HW, HW, Ci, Ci, HW, HW, Co, ld, 3, 3, 1, 1, code, hooks=hk), n=10)
        print(f"wgrad 3x3s1 {Ci:3d}->{Co:3d} @{HW:3d}^2  blocks {b:5d}: {us:8.1f} us (min {mn:8.1f})  {flops / us / 1e6:7.1f} TF", flush=True)
    del x, dy, dwp

print("slab form (cn_conv2d_wgrad_direct: private slabs + one reduction launch into the parameter layout)")
for (N, HW, Ci, Co) in SHAPES:
    ld = (Co + 15) // 16 * 16
    x = torch.randn(N, HW, HW, Ci, device=DEV).to(dt)
    dy = torch.randn(N, HW, HW, ld, device=DEV).to(dt)
    dw = torch.zeros(Co, Ci, 3, 3, device=DEV)
    flops = 2.0 * N * HW * HW * Ci * Co * 9
    for b in blocks:
        hk = _hip.Hooks().set(wgrad_blocks=b)
        n = int(_hip.query("cn_conv2d_wgrad_direct_bytes_h", N, HW, HW, Ci, Ci, HW, HW, Co, ld, 3, 3, 1, 1, code, b))
        if not n:
            continue
        ws = torch.empty(n, dtype=torch.uint8, device=DEV)
        us, mn = timeit(lambda: _hip.call("cn_conv2d_wgrad_direct", x, dy, dw, None, 1, ws, n, N, HW, HW, Ci, Ci, HW, HW, Co, ld, 3, 3, 1, 1, code, hooks=hk), n=10)
        print(f"wgrad 3x3s1 {Ci:3d}->{Co:3d} @{HW:3d}^2  blocks {b:5d}: {us:8.1f} us (min {mn:8.1f})  {flops / us / 1e6:7.1f} TF   slabs {n / 1e6:6.1f} MB", flush=True)
    del x, dy, dw

print("stride 2 (input size given): cn_conv2d_wgrad (generic split-K kernel, one workgroup column per tap) vs cn_conv2d_wgrad_direct (halo-tile slab form)")
for (N, HW, Ci, Co) in [(64, 256, 32, 64), (64, 128, 64, 128), (64, 64, 128, 256), (64, 32, 256, 512)]:
    OHW = HW // 2
    x = torch.randn(N, HW, HW, Ci, device=DEV).to(dt)
    dy = torch.randn(N, OHW, OHW, Co, device=DEV).to(dt)
    dwp = torch.zeros((Co + 31) // 32 * 32, 9 * Ci, device=DEV)
    dw = torch.zeros(Co, Ci, 3, 3, device=DEV)
    flops = 2.0 * N * OHW * OHW * Ci * Co * 9
    for b in blocks:
        hk = _hip.Hooks().set(wgrad_blocks=b)
        us, mn = timeit(lambda: _hip.call("cn_conv2d_wgrad", x, dy, dwp, None, N, HW, HW, Ci, Ci, OHW, OHW, Co, Co, 3, 3, 2, 1, code, hooks=hk), n=10)
        n = int(_hip.query("cn_conv2d_wgrad_direct_bytes_h", N, HW, HW, Ci, Ci, OHW, OHW, Co, Co, 3, 3, 2, 1, code, b))
        ws = torch.empty(max(n, 16), dtype=torch.uint8, device=DEV)
        us2, mn2 = timeit(lambda: _hip.call("cn_conv2d_wgrad_direct", x, dy, dw, None, 1, ws, n, N, HW, HW, Ci, Ci, OHW, OHW, Co, Co, 3, 3, 2, 1, code, hooks=hk), n=10) if n else (float("nan"), 0)
        print(f"wgrad 3x3s2 {Ci:3d}->{Co:3d} @{HW:3d}^2  blocks {b:5d}: generic {us:8.1f} us   slab {us2:8.1f} us  ({flops / us2 / 1e6:6.1f} TF)", flush=True)
    del x, dy, dwp, dw
