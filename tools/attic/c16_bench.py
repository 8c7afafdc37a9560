"""Level0 / level1 convolutions of DLA-34 at batch 64 (16 input channels, 512^2): forward 16->16, its data gradient (mirrored taps),
forward 16->32 stride 2.  A/B: CN_DISABLE_CONV_C16R=1 (read once per process) puts them back on the strip / implicit-GEMM kernels.
usage: python tools/attic/c16_bench.py [N]"""
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
    img = torch.randn(N, 3, 512, 512, device=DEV)
    ws = (torch.randn(16, 3, 7, 7, generator=g) * 0.1).to(DEV)
    ys = torch.empty(N, 512, 512, 16, device=DEV, dtype=dt)
    us = timeit(lambda: _hip.call("cn_stem_conv_fwd", img, ws, None, None, ys, N, 3, 512, 512, 16, 7, 7, 1, 3, 512, 512, 0, _hip.dtype_code(dt)))
    print("stem 7x7 3->16 @512^2 fwd         ", mb(img.numel() * 4 + ys.numel() * 2, us), "   CN_DISABLE_STEM_ROWS =", os.environ.get("CN_DISABLE_STEM_ROWS"))
    if not os.environ.get("CN_DISABLE_CONV_C16R"):
        def aff(wp_, Co, s, OH):
            return ops._igemm(x, wp_, None, None, Co, 3, 3, s, 1, False, False, OH, OH, pre=(ss, True))
        us = timeit(lambda: aff(wp, 16, 1, 512))
        print("conv3x3 16->16 @512^2 fwd  + BN/ReLU on load", mb(2 * x.numel() * 2, us))
        us = timeit(lambda: aff(wp2, 32, 2, 256))
        print("conv3x3 16->32 s2 fwd      + BN/ReLU on load", mb(x.numel() * 2 + N * 256 * 256 * 32 * 2, us))


def tail():
    """the last kernels of a step: BN backward of the stem's BN + the stem's weight gradient, unfused vs fused (cn_stem_conv_wgrad_bn)"""
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dt = torch.bfloat16
    code = _hip.dtype_code(dt)
    img = torch.randn(N, 3, 512, 512, device=DEV)
    y = torch.randn(N, 512, 512, 16, device=DEV).to(dt)
    dy = torch.randn(N, 512, 512, 16, device=DEV).to(dt)
    dx = torch.empty_like(dy)
    gamma = torch.rand(16, device=DEV) + 0.5
    stats = torch.stack([torch.zeros(16, device=DEV), torch.ones(16, device=DEV), gamma, torch.zeros(16, device=DEV)]).contiguous()
    dw = torch.zeros(16, 3, 7, 7, device=DEV)
    dg, db = torch.zeros(16, device=DEV), torch.zeros(16, device=DEV)
    sink = torch.zeros(_hip.query("cn_bn_stats_slots"), 2, 16, device=DEV)
    coef = torch.zeros(5, 16, device=DEV)
    npix = N * 512 * 512
    for blocks in (160, 1536):
        hk = _hip.Hooks().set(wgrad_blocks=blocks)
        us = timeit(lambda: _hip.call("cn_stem_conv_wgrad", img, dy, dw, N, 3, 512, 512, 16, 7, 7, 1, 3, 512, 512, code, hooks=hk))
        print(f"stem wgrad              [wgrad blocks {blocks:4d}] {us:8.1f} us")
        us = timeit(lambda: _hip.call("cn_stem_conv_wgrad_bn", img, dy, y, coef, dw, N, 3, 512, 512, 16, 7, 7, 1, 3, 512, 512, 1, code, hooks=hk))
        print(f"stem wgrad through BN   [wgrad blocks {blocks:4d}] {us:8.1f} us")
    def bwd():
        sink.zero_()
        _hip.call("cn_bn_train_bwd_sink", dy, y, None, gamma, stats[0], stats[1], stats[2:], dx, None, None, dg, db, 1, sink, sink.shape[0], None, 0,
                  npix, 16, 1, code)
    us = timeit(bwd)
    print(f"cn_bn_train_bwd_sink 16 ch (+ zero of the sink) {us:8.1f} us")
    def st():
        sink.zero_()
        _hip.call("cn_bn_bwd_stats", dy, y, None, stats[0], stats[1], stats[2:], sink, sink.shape[0], npix, 16, 1, code)
        _hip.call("cn_bn_bwd_coef_sink", sink, sink.shape[0], gamma, stats[0], stats[1], stats[2:], dg, db, 1, coef, None, 0, npix, 16)
    us = timeit(st)
    print(f"cn_bn_bwd_stats + cn_bn_bwd_coef_sink (+ zero)  {us:8.1f} us")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "tail":
        tail()
        sys.exit(0)
    main()
