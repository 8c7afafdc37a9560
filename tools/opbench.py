"""Isolated timings of single entry points at the bench's shapes (HIP events on the launch stream, median of N):
    python tools/opbench.py decode bn            # families: decode bn
Prints microseconds and the achieved GB/s against the algorithmic bytes of each call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centernet_amd import _hip, ops  # noqa: E402

DEV = "cuda"


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def report(name, us, mn, nbytes):
    print(f"{name:58s} {us:9.1f} us (min {mn:8.1f})  {nbytes / us / 1e3:8.1f} GB/s", flush=True)


def bench_decode():
    from centernet_amd.decode.ctdet import ctdet_decode
    from centernet_amd.decode.multi_pose import multi_pose_decode
    B, C, H, W = 64, 80, 128, 128
    g = torch.Generator(device="cpu").manual_seed(3)
    maps = {"realistic sigmoid(0.5 N - 2.19)": torch.sigmoid(torch.randn(B, C, H, W, generator=g) * 0.5 - 2.19),
            "bf16 logits of an untrained net": torch.sigmoid((torch.randn(B, C, H, W, generator=g) * 0.01 - 2.19).bfloat16().float()),
            "flat 0.25": torch.full((B, C, H, W), 0.25)}
    wh = torch.rand(B, 2, H, W, generator=g).to(DEV) * 30
    reg = torch.rand(B, 2, H, W, generator=g).to(DEV)
    for tag, heat in maps.items():
        heat = heat.to(DEV)
        us, mn = timeit(lambda: ctdet_decode(heat, wh, reg, K=100))
        report(f"cn_ctdet_decode B64 C80 128^2 [{tag}]", us, mn, heat.numel() * 4)
        s = torch.empty(B, C, 100, device=DEV); i = torch.empty(B, C, 100, dtype=torch.int32, device=DEV)
        us, mn = timeit(lambda: _hip.call("cn_topk_channel", heat, s, i, B, C, H, W, 100, 1))
        report(f"  cn_topk_channel (stage 1 only)", us, mn, heat.numel() * 4)
    B = 32
    d = lambda *sh: torch.rand(*sh, generator=g).to(DEV)
    heat, hp = torch.sigmoid(d(B, 1, H, W) * 4 - 4), torch.sigmoid(d(B, 17, H, W) * 4 - 4)
    whp, kps, rg = d(B, 2, H, W), d(B, 34, H, W), reg[:B].contiguous()
    us, mn = timeit(lambda: multi_pose_decode(heat, whp, kps, reg=rg, hm_hp=hp, hp_offset=rg, K=100))
    report("cn_multi_pose_decode B32 J17 128^2", us, mn, (1 + 17 + 34 + 6) * B * H * W * 4)


def bench_bn():
    dt = torch.bfloat16
    for npix, C in [(64 * 128 * 128, 64), (64 * 512 * 512, 16), (64 * 256 * 256, 32), (64 * 64 * 64, 128), (64 * 32 * 32, 256),
                    (64 * 16 * 16, 512)]:
        x = torch.randn(npix, C, device=DEV).to(dt)
        dy = torch.randn(npix, C, device=DEV).to(dt)
        res = torch.randn(npix, C, device=DEV).to(dt)
        y, dx, dres = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        gamma, beta = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        stats = torch.empty(4, C, device=DEV)
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        ws, n = ops._bn_ws(npix, C, DEV)
        code = _hip.dtype_code(dt)
        nb = npix * C * 2

        def fwd(r):
            _hip.call("cn_bn_train_fwd", x, r, y, gamma, beta, rm, rv, stats[0], stats[1], stats[2:], npix, C, 0.1, 1e-5, 1, code, ws, n)

        def bwd(yy, dr):
            _hip.call("cn_bn_train_bwd", dy, x, yy, gamma, stats[0], stats[1], stats[2:], dx, dr, dg, db, 1, npix, C, 1, code, ws, n)
        us, mn = timeit(lambda: fwd(None)); report(f"cn_bn_train_fwd npix={npix} C={C} relu (3 passes)", us, mn, 3 * nb)
        us, mn = timeit(lambda: fwd(res)); report(f"cn_bn_train_fwd npix={npix} C={C} +res relu (4 passes)", us, mn, 4 * nb)
        us, mn = timeit(lambda: bwd(None, None)); report(f"cn_bn_train_bwd npix={npix} C={C} mask from x (5 passes)", us, mn, 5 * nb)
        us, mn = timeit(lambda: bwd(y, dres)); report(f"cn_bn_train_bwd npix={npix} C={C} +res, mask from y (8 passes)", us, mn, 8 * nb)
        del x, dy, res, y, dx, dres


def bench_dcn():
    """the four DCNv2 kernels at the DLA-34 up-path shapes (batch 64), zero offsets (an untrained net) and N(0, 0.5 px) offsets"""
    dt = torch.bfloat16
    code = _hip.dtype_code(dt)
    shapes = [(128, 64, 64), (64, 128, 64), (64, 128, 128), (32, 256, 128), (32, 256, 256), (16, 512, 256)]
    if os.environ.get("DCN_SHAPES"):
        shapes = shapes[:int(os.environ["DCN_SHAPES"])]
    for HW, Ci, Co in shapes:
        N = H = W = None
        N, H, W = 64, HW, HW
        for tag, sigma in (("zero offsets", 0.0), ("offsets N(0,0.5)", 0.5)):
            g = torch.Generator(device="cpu").manual_seed(5)
            x = torch.randn(N, H, W, Ci, device=DEV).to(dt)
            om = torch.zeros(N, H, W, 32, device=DEV)
            if sigma:
                om[..., :18] = torch.randn(N, H, W, 18, device=DEV) * sigma
                om[..., 18:27] = torch.randn(N, H, W, 9, device=DEV)
            w = (torch.randn(Co, Ci, 3, 3, generator=g) * (2.0 / (9 * Ci)) ** 0.5).to(DEV)
            bias = torch.zeros(Co, device=DEV)
            wp1, wp0, wp2 = ops.pack_weight(w, 1, dt), ops.pack_weight(w, 0, dt), ops.pack_weight(w, 2, dt)
            y = torch.empty(N, H, W, Co, device=DEV, dtype=dt)
            dy = torch.randn(N, H, W, Co, device=DEV).to(dt)
            flops = 2.0 * N * H * W * 9 * Ci * Co
            us, mn = timeit(lambda: _hip.call("cn_dcn_fwd", x, om, wp1, bias, y, N, H, W, Ci, Ci, Co, Co, 32, 0, code), n=10)
            print(f"dcn {Ci:3d}->{Co:3d} @{HW:3d}^2 [{tag:16s}] fwd   {us:8.1f} us  {flops / us / 1e6:7.1f} TF", flush=True)
            far = ops._far_buffer((N, H, W, Ci), DEV)
            flag = torch.zeros(1, dtype=torch.int32, device=DEV)
            slabs = _hip.query("cn_dcn_bwd_dom_slabs", Ci, Co, code)
            dom = torch.empty((max(slabs, 1), N, H, W, 32), device=DEV)
            if slabs != Ci // 64:
                slabs, dom = 1, torch.zeros(1, N, H, W, 32, device=DEV)
            us, mn = timeit(lambda: _hip.call("cn_dcn_bwd_dom", dy, wp2, x, om, dom, slabs, far, flag, N, H, W, Ci, Co, Co, Ci, 32, code), n=10)
            print(f"dcn {Ci:3d}->{Co:3d} @{HW:3d}^2 [{tag:16s}] dom   {us:8.1f} us  {flops / us / 1e6:7.1f} TF", flush=True)
            dx = torch.empty_like(x)
            us, mn = timeit(lambda: _hip.call("cn_dcn_bwd_dx", dy, wp0, om, far, flag, dx, N, H, W, Ci, Co, 32, code), n=10)
            print(f"dcn {Ci:3d}->{Co:3d} @{HW:3d}^2 [{tag:16s}] dx    {us:8.1f} us  {flops / us / 1e6:7.1f} TF", flush=True)
            dwp = torch.zeros(ops.rup(Co, 32), 9 * Ci, device=DEV)
            us, mn = timeit(lambda: _hip.call("cn_dcn_wgrad", x, om, dy, dwp, N, H, W, Ci, Ci, Co, Co, 32, code), n=10)
            print(f"dcn {Ci:3d}->{Co:3d} @{HW:3d}^2 [{tag:16s}] wgrad {us:8.1f} us  {flops / us / 1e6:7.1f} TF", flush=True)
            del x, om, y, dy, dx, dom


def bench_conv():
    """3x3 / stride 1 convolutions (forward entry point; data gradients are the same kernels with mirrored taps) at the DLA-34
    shapes, batch 64, random N(0,1) activations.  Each line: this build, then CN_DISABLE-style A/B is done by running the script
    again with CN_DISABLE_CONV_WS=1 (the switch is read once per process)."""
    dt = torch.bfloat16
    shapes = [(512, 16, 16, False), (128, 64, 256, True), (128, 64, 64, False), (128, 64, 27, False), (128, 256, 64, False), (64, 128, 128, False),
              (32, 256, 256, False), (16, 512, 512, False), (64, 64, 64, False)]
    for HW, Ci, Co, relu in shapes:
        N = 64
        g = torch.Generator(device="cpu").manual_seed(7)
        x = torch.randn(N, HW, HW, Ci, device=DEV).to(dt)
        w = (torch.randn(Co, Ci, 3, 3, generator=g) * (2.0 / (9 * Ci)) ** 0.5).to(DEV)
        wp = ops.pack_weight(w, 1, dt)
        bias = torch.zeros(Co, device=DEV)
        flops = 2.0 * N * HW * HW * 9 * Ci * Co
        us, mn = timeit(lambda: ops._igemm(x, wp, bias, None, Co, 3, 3, 1, 1, False, relu, HW, HW), n=15)
        print(f"conv3x3 {Ci:3d}->{Co:3d} @{HW:3d}^2 bs64  {us:8.1f} us (min {mn:8.1f})  {flops / us / 1e6:7.1f} TF  {flops / us / 2.5e9 * 100:5.1f} % of MFMA peak", flush=True)
        del x


def bench_conv1x1():
    """1x1 / stride 1 convolutions of DLA-34 at batch 64 (forward entry point; the data gradients are 1x1 convs with the other
    packing).  A/B: CN_DISABLE_CONV1X1_STREAM=1 (read once per process) puts them back on the implicit-GEMM kernel."""
    dt = torch.bfloat16
    for HW, Ci, Co in [(128, 256, 80), (128, 256, 2), (64, 128, 128), (128, 64, 64), (32, 256, 256), (64, 64, 128), (128, 32, 64),
                       (64, 128, 64), (128, 64, 32), (32, 128, 256), (32, 256, 128)]:
        N = 64
        g = torch.Generator(device="cpu").manual_seed(9)
        x = torch.randn(N, HW, HW, Ci, device=DEV).to(dt)
        w = (torch.randn(Co, Ci, 1, 1, generator=g) * (2.0 / Ci) ** 0.5).to(DEV)
        wp = ops.pack_weight(w, 1, dt)
        nbytes = N * HW * HW * (Ci + ops.rup(Co, 16)) * 2
        us, mn = timeit(lambda: ops._igemm(x, wp, None, None, Co, 1, 1, 1, 0, False, False, HW, HW), n=15)
        report(f"conv1x1 {Ci:3d}->{Co:3d} @{HW:3d}^2 bs64", us, mn, nbytes)
        del x
    # data gradients of the heads' output convs (256 -> C): dy [P][rup16(C)] -> dx [P][256] masked by the hidden activation
    for C in (2, 80):
        N, HW, Ci = 64, 128, 256
        ld = ops.rup(C, 16)
        dy = torch.randn(N, HW, HW, ld, device=DEV).to(dt)
        h = torch.randn(N, HW, HW, Ci, device=DEV).to(dt)
        w = torch.randn(C, Ci, 1, 1, device=DEV) * 0.05
        wpd = ops.pack_weight(w, 0, dt)
        nbytes = N * HW * HW * (ld + 2 * Ci) * 2
        us, mn = timeit(lambda: ops._igemm(dy, wpd, None, h, Ci, 1, 1, 1, 0, True, 2, HW, HW), n=15)
        report(f"conv1x1 dgrad {C:2d}->256 @128^2 bs64 via cn_conv2d_fwd", us, mn, nbytes)
        dx = torch.empty_like(h)
        us, mn = timeit(lambda: _hip.call("cn_conv1x1_smallk", dy, wpd, h, dx, N * HW * HW, C, ld, Ci, Ci, Ci, 2, _hip.dtype_code(dt)), n=15)
        report(f"conv1x1 dgrad {C:2d}->256 @128^2 bs64 via cn_conv1x1_smallk", us, mn, nbytes)


if __name__ == "__main__":
    fams = sys.argv[1:] or ["decode", "bn"]
    print("CN_DISABLE_TOPK_STREAM =", os.environ.get("CN_DISABLE_TOPK_STREAM"))
    for f in fams:
        globals()["bench_" + f]()
