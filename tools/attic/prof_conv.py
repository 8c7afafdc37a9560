"""Run a few shapes of the conv / DCN kernels in isolation (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import ops, nn as hnn

dev = "cuda"
dt = torch.bfloat16
N = 64
def conv(ci, co, hw, reps=3):
    x = torch.randn(N, hw, hw, ci, device=dev).to(dt)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    wp = ops.pack_weight(w, 1, dt)
    for _ in range(reps):
        y = ops._igemm(x, wp, None, None, co, 3, 3, 1, 1, False, False, hw, hw)
    torch.cuda.synchronize()
    return y
which = sys.argv[1] if len(sys.argv) > 1 else "conv"
if which == "conv":
    conv(64, 256, 128); conv(256, 64, 128); conv(64, 64, 128); conv(128, 128, 64); conv(256, 256, 32)
elif which == "head":
    def conv1(ci, co, hw, reps=3):
        x = torch.randn(N, hw, hw, ci, device=dev).to(dt)
        w = torch.randn(co, ci, 1, 1, device=dev) * 0.05
        wp = ops.pack_weight(w, 1, dt)
        for _ in range(reps):
            y = ops._igemm(x, wp, None, None, co, 1, 1, 1, 0, False, False, hw, hw)
        torch.cuda.synchronize()
    conv1(16, 256, 128); conv1(80, 256, 128); conv1(256, 80, 128); conv1(128, 64, 128); conv1(320, 128, 64)
elif which == "headbwd":
    hw = 128
    for ci in (16, 80):
        dy = torch.randn(N, hw, hw, ci, device=dev).to(dt)
        hid = torch.randn(N, hw, hw, 256, device=dev).to(dt)
        w = torch.randn(ci, 256, 1, 1, device=dev) * 0.05            # forward weight [Co=ci][Ci=256]
        wpd = ops.pack_weight(w, 0, dt)
        for mode in (2, 0, 2):
            dx = ops._igemm(dy, wpd, None, hid if mode == 2 else None, 256, 1, 1, 1, 0, True, mode, hw, hw)
    torch.cuda.synchronize()
elif which == "topk":
    from centernet_amd._hip import call
    heat = torch.sigmoid(torch.randn(64, 80, 128, 128, device=dev) * 0.5 - 2.19)
    sc = torch.empty(64, 80, 100, device=dev); ind = torch.empty(64, 80, 100, dtype=torch.int32, device=dev)
    for nms in (1, 0, 1):
        call("cn_topk_channel", heat, sc, ind, 64, 80, 128, 128, 100, nms)
    nm = torch.empty_like(heat)
    call("cn_nms3x3", heat, nm, 64, 80, 128, 128)
    call("cn_topk_channel", nm, sc, ind, 64, 80, 128, 128, 100, 0)
    torch.cuda.synchronize()
elif which == "bn":
    for (npix, C) in ((1 << 20, 64), (1 << 24, 16), (1 << 18, 128)):
        bn = hnn.BatchNorm2d(C).to(dev)
        x = torch.randn(npix // 1024, 32, 32, C, device=dev).to(dt).requires_grad_(True)
        for _ in range(3):
            y = bn(x, None, True)
            y.backward(torch.randn_like(y))
    torch.cuda.synchronize()
elif which == "dcn2":
    for (ci, co, hw) in ((128, 64, 64), (128, 128, 64), (256, 128, 32), (64, 64, 128)):
        m = hnn.DCN(ci, co).to(dev)
        torch.nn.init.normal_(m.conv_offset_mask.weight, std=0.01)
        x = torch.randn(N, hw, hw, ci, device=dev).to(dt).requires_grad_(True)
        for _ in range(2):
            y = m(x); y.backward(torch.randn_like(y))
    torch.cuda.synchronize()
elif which == "dcn":
    m = hnn.DCN(64, 64).to(dev)
    torch.nn.init.normal_(m.conv_offset_mask.weight, std=0.01)
    x = torch.randn(N, 128, 128, 64, device=dev).to(dt).requires_grad_(True)
    for _ in range(2):
        y = m(x); y.backward(torch.randn_like(y))
    torch.cuda.synchronize()
