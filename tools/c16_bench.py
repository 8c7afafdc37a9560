"""Level0 / level1 convolutions of DLA-34 at batch 64 (16 input channels, 512^2): forward 16->16, its data gradient (mirrored taps),
forward 16->32 stride 2.  A/B: CN_DISABLE_CONV_C16R=1 (read once per process) puts them back on the strip / implicit-GEMM kernels.
usage: python tools/c16_bench.py [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centernet_amd import _hip, ops  # noqa: E402

DEV = torch.device("cuda:0")


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dt = torch.bfloat16
    print("CN_DISABLE_CONV_C16R =", os.environ.get("CN_DISABLE_CONV_C16R"))
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn(N, 512, 512, 16, device=DEV).to(dt)
    w16 = (torch.randn(16, 16, 3, 3, generator=g) * (2.0 / 144) ** 0.5).to(DEV)
    w32 = (torch.randn(32, 16, 3, 3, generator=g) * (2.0 / 144) ** 0.5).to(DEV)
    ss = torch.cat([torch.rand(16, device=DEV) + 0.5, torch.randn(16, device=DEV) * 0.1]).contiguous()
    mb = lambda nb, us: f"{us:8.1f} us  {nb / us / 1e6:6.2f} TB/s"
    wp = ops.pack_weight(w16, 1, dt)
    us = timeit(lambda: ops._igemm(x, wp, None, None, 16, 3, 3, 1, 1, False, False, 512, 512))
    print("conv3x3 16->16 @512^2 fwd        ", mb(2 * x.numel() * 2, us))
    wpd = ops.pack_weight(w16, 0, dt)
    us = timeit(lambda: ops._igemm(x, wpd, None, None, 16, 3, 3, 1, 1, True, False, 512, 512))
    print("conv3x3 16->16 @512^2 dgrad      ", mb(2 * x.numel() * 2, us))
    wp2 = ops.pack_weight(w32, 1, dt)
    us = timeit(lambda: ops._igemm(x, wp2, None, None, 32, 3, 3, 2, 1, False, False, 256, 256))
    print("conv3x3 16->32 s2 @512^2 fwd     ", mb(x.numel() * 2 + N * 256 * 256 * 32 * 2, us))
    if not os.environ.get("CN_DISABLE_CONV_C16R"):
        def aff(wp_, Co, s, OH):
            return ops._igemm(x, wp_, None, None, Co, 3, 3, s, 1, False, False, OH, OH, pre=(ss, True))
        us = timeit(lambda: aff(wp, 16, 1, 512))
        print("conv3x3 16->16 @512^2 fwd  + BN/ReLU on load", mb(2 * x.numel() * 2, us))
        us = timeit(lambda: aff(wp2, 32, 2, 256))
        print("conv3x3 16->32 s2 fwd      + BN/ReLU on load", mb(x.numel() * 2 + N * 256 * 256 * 32 * 2, us))


if __name__ == "__main__":
    main()
