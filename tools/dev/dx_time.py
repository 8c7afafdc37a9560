"""Isolated timing of cn_dcn_bwd_dx at the DLA-34 shapes handled by dcn_dx_bm_kernel.  python tools/dev/dx_time.py [tag]   (CN_LIB_PATH selects the library)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import opbench
from centernet_amd import _hip, ops
tag = sys.argv[1] if len(sys.argv) > 1 else ""
dt = torch.bfloat16; code = _hip.dtype_code(dt)
for HW, Ci, Co in [(128, 64, 64), (64, 128, 64), (32, 256, 64)]:
    N, H, W = 64, HW, HW
    for otag, sigma in (("zero", 0.0), ("N(0,0.5)", 0.5)):
        om = torch.zeros(N, H, W, 32, device="cuda")
        if sigma:
            om[..., :18] = torch.randn(N, H, W, 18, device="cuda") * sigma
            om[..., 18:27] = torch.randn(N, H, W, 9, device="cuda")
        w = (torch.randn(Co, Ci, 3, 3) * (2.0 / (9 * Ci)) ** 0.5).cuda()
        wp0 = ops.pack_weight(w, 0, dt)
        dy = torch.randn(N, H, W, Co, device="cuda").to(dt)
        far = ops._far_buffer((N, H, W, Ci), "cuda")
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        dx = torch.empty(N, H, W, Ci, device="cuda", dtype=dt)
        us, mn = opbench.timeit(lambda: _hip.call("cn_dcn_bwd_dx", dy, wp0, om, far, flag, dx, N, H, W, Ci, Co, 32, code), n=20)
        print(f"{tag:6s} dcn dx {Ci:3d}<-{Co:3d} @{HW:3d}^2 [{otag:9s}] {us:8.1f} us (min {mn:8.1f})", flush=True)
