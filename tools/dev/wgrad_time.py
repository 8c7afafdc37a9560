"""Isolated timing of cn_dcn_wgrad at the DLA-34 shapes, at the side-stream grid (160) and the full grid (256); env switches select the variant.
    python tools/dev/wgrad_time.py [tag]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import opbench
from centernet_amd import _hip, ops
tag = sys.argv[1] if len(sys.argv) > 1 else ""
dt = torch.bfloat16; code = _hip.dtype_code(dt)
for HW, Ci, Co in [(128, 64, 64), (64, 128, 64), (64, 128, 128)]:
    N, H, W = 64, HW, HW
    for otag, sigma in (("zero", 0.0), ("N(0,0.5)", 0.5)):
        x = torch.randn(N, H, W, Ci, device="cuda").to(dt)
        om = torch.zeros(N, H, W, 32, device="cuda")
        if sigma:
            om[..., :18] = torch.randn(N, H, W, 18, device="cuda") * sigma
            om[..., 18:27] = torch.randn(N, H, W, 9, device="cuda")
        dy = torch.randn(N, H, W, Co, device="cuda").to(dt)
        dwp = torch.zeros(ops.rup(Co, 32), 9 * Ci, device="cuda")
        for blocks in (160, 256):
            h = ops.Hooks().set(wgrad_blocks=blocks)
            us, mn = opbench.timeit(lambda: _hip.call("cn_dcn_wgrad", x, om, dy, dwp, N, H, W, Ci, Ci, Co, Co, 32, code, hooks=h), n=15)
            print(f"{tag:8s} dcn wgrad {Ci:3d}->{Co:3d} @{HW:3d}^2 [{otag:9s}] blocks {blocks}: {us:8.1f} us (min {mn:8.1f})", flush=True)
