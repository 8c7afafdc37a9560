"""One entry point of the 64->64 @128^2 batch-64 DCNv2 layer a few times (for rocprofv3 counter passes): python tools/attic/gs_time.py fwd|dom|dw [sigma]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centernet_amd import _hip, ops  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
N, H, W, Ci, Co = 64, 128, 128, 64, 64
dt = torch.bfloat16
code = _hip.dtype_code(dt)
x = torch.randn(N, H, W, Ci, device="cuda").to(dt)
om = torch.zeros(N, H, W, 32, device="cuda")
if sigma:
    om[..., :18] = torch.randn(N, H, W, 18, device="cuda") * sigma
    om[..., 18:27] = torch.randn(N, H, W, 9, device="cuda")
w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.04
bias = torch.zeros(Co, device="cuda")
y = torch.empty(N, H, W, Co, device="cuda", dtype=dt)
dy = torch.randn(N, H, W, Co, device="cuda").to(dt)
if what == "fwd":
    wp = ops.pack_weight(w, 1, dt)
    run = lambda: _hip.call("cn_dcn_fwd", x, om, wp, bias, y, N, H, W, Ci, Ci, Co, Co, 32, 0, code)
elif what == "dom":
    wp = ops.pack_weight(w, 2, dt)
    dom = torch.empty(N, H, W, 32, device="cuda", dtype=dt)
    far = torch.zeros(N, H, W, Ci, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    run = lambda: _hip.call("cn_dcn_bwd_dom", dy, wp, x, om, dom, 0, far, flag, N, H, W, Ci, Co, Co, Ci, 32, code)
else:
    dwp = torch.zeros(Co, 9 * Ci, device="cuda")
    run = lambda: _hip.call("cn_dcn_wgrad", x, om, dy, dwp, N, H, W, Ci, Ci, Co, Co, 32, code)
for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record()
torch.cuda.synchronize()
print(f"{what} sigma {sigma}: {e0.elapsed_time(e1) * 200:.1f} us per launch")
