"""Isolated timing of the two 512x512 weight-gradient launches that end a step (stem 7x7 3->16 on the fp32 NCHW image, level0 3x3
16->16) at full width: python tools/attic/stem_wgrad_bench.py   (CN_LIB_PATH: another build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import _hip
from centernet_amd._hip import call
dev = "cuda"
N, H, W = 64, 512, 512
img = torch.rand(N, 3, H, W, device=dev)
dy = torch.randn(N, H, W, 16, device=dev).bfloat16()
x16 = torch.randn(N, H, W, 16, device=dev).bfloat16()
dw = torch.zeros(16, 3, 7, 7, device=dev)
dwp = torch.zeros(32, 9 * 16, device=dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


t = timed(lambda: call("cn_stem_conv_wgrad", img, dy, dw, N, 3, H, W, 16, 7, 7, 1, 3, H, W, _hip.CN_BF16))
print(f"stem 7x7 3->16 wgrad    {t:7.1f} us   ({(img.numel() * 4 + dy.numel() * 2) / t / 1e6:.2f} TB/s algorithmic)")
t = timed(lambda: call("cn_conv2d_wgrad", x16, dy, dwp, None, N, H, W, 16, 16, H, W, 16, 16, 3, 3, 1, 1, _hip.CN_BF16))
print(f"level0 3x3 16->16 wgrad {t:7.1f} us   ({(x16.numel() * 2 + dy.numel() * 2) / t / 1e6:.2f} TB/s algorithmic)")
