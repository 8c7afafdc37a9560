"""GPU parity of every HIP operator (through the C ABI) against plain torch-CPU fp32 references of the same op
(the oracle's building blocks).  fp32 compute: tight tolerances; bf16 compute: reference built from
bf16-rounded operands with fp32 accumulation."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from centernet_amd import rng

pytestmark = pytest.mark.gpu

DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def ops():
    from centernet_amd import ops as o
    return o


def to_nhwc(x, dt):
    return x.detach().permute(0, 2, 3, 1).contiguous().to(dt).to(DEV)


def to_nchw(y):
    return y.detach().float().cpu().permute(0, 3, 1, 2).contiguous()


def rnd(x, dt):
    """operand as the kernel sees it"""
    return x.detach().to(dt).float().clone()


def close(a, b, dt, what, scale=None):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    s = float(b.abs().max()) if scale is None else scale
    s = max(s, 1e-6)
    tol = 2e-5 if dt == torch.float32 else 1.5e-2
    err = float((a - b).abs().max()) / s
    assert err < tol, f"{what}: max err / max|ref| = {err:.3e} (tol {tol})"


CONVS = [  # N,H,W,Ci,Co,k,s,p,bias,relu
    (2, 16, 16, 16, 16, 3, 1, 1, False, False),
    (2, 9, 70, 16, 16, 3, 1, 1, True, True),       # direct 16-channel kernel, ragged row strips
    (1, 33, 129, 16, 16, 3, 1, 1, False, False),
    (1, 70, 45, 16, 16, 3, 1, 1, True, False),      # row-walking 16-channel kernel: several ring rounds per wave, ragged last round
    (1, 75, 45, 16, 32, 3, 2, 1, True, True),       # ... stride 2 (level1 forward), odd sizes
    (1, 64, 200, 16, 32, 3, 2, 1, False, False),    # ... wide enough for INTERIOR waves (the guard-free instantiation of the row loop)
    (1, 52, 130, 16, 16, 3, 1, 1, False, False),
    (2, 17, 19, 32, 48, 3, 1, 1, True, True),
    (1, 16, 16, 64, 128, 3, 2, 1, False, False),
    (2, 38, 70, 16, 32, 3, 2, 1, False, False),     # level1 shape: its data gradient runs on dgrad_s2_c32to16_kernel<1,1>
    (2, 21, 37, 32, 64, 3, 2, 1, False, False),     # level2 shape: dgrad_s2_c32to16_kernel<2,2> (odd sizes: ragged strips)
    (1, 15, 13, 64, 64, 3, 2, 1, False, False),
    (2, 8, 8, 128, 256, 1, 1, 0, False, False),
    (2, 16, 16, 64, 128, 1, 2, 0, False, False),
    (2, 12, 12, 256, 27, 3, 1, 1, True, False),
    (1, 9, 9, 256, 2, 1, 1, 0, True, False),
    (2, 13, 11, 64, 1, 1, 1, 0, True, False),      # 1- / 2- / 4-channel heads: data gradient on cn_conv1x1_smallk (bf16)
    (1, 7, 19, 512, 4, 1, 1, 0, False, False),
    (2, 4, 4, 512, 256, 3, 1, 1, False, False),
    (1, 24, 24, 448, 128, 1, 1, 0, False, False),
    (2, 32, 32, 256, 256, 3, 1, 1, False, False),   # Hourglass shapes (384-wide levels, 2x2 / 1x1 maps, strided residuals)
    (2, 16, 16, 256, 384, 3, 2, 1, False, False),
    (2, 8, 8, 384, 384, 3, 1, 1, False, False),
    (2, 2, 2, 384, 512, 3, 2, 1, False, False),
    (2, 1, 1, 512, 512, 3, 1, 1, False, False),
    (2, 4, 4, 384, 512, 1, 2, 0, False, False),
    (2, 32, 32, 128, 256, 3, 2, 1, False, False),
]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", CONVS)
def test_conv2d_fwd_bwd(cfg, dt):
    N, H, W, Ci, Co, k, s, p, bias, relu = cfg
    x = rng.t_normal(1, f"x{cfg}", (N, Ci, H, W))
    w = rng.t_normal(1, f"w{cfg}", (Co, Ci, k, k), 0, (2.0 / (Ci * k * k)) ** 0.5)
    b = rng.t_normal(1, f"b{cfg}", (Co,), 0, 0.1) if bias else None
    xr, wr = rnd(x, dt).requires_grad_(True), rnd(w, dt).requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, s, p)
    if relu:
        yr = F.relu(yr)
    gy = rng.t_normal(1, f"g{cfg}", tuple(yr.shape))
    yr.backward(rnd(gy, dt))

    xg = to_nhwc(x, dt).requires_grad_(True)
    wg = w.to(DEV).requires_grad_(True)
    bg = b.to(DEV).requires_grad_(True) if bias else None
    y = ops().conv2d(xg, wg, bg, s, p, relu)
    assert y.shape[-1] == (Co + 15) // 16 * 16
    close(to_nchw(y)[:, :Co], yr, dt, "conv fwd")
    if y.shape[-1] != Co:
        assert float(y[..., Co:].abs().max()) == 0.0, "channel padding must stay zero"
    gyp = torch.zeros(y.shape, dtype=dt, device=DEV)
    gyp[..., :Co] = to_nhwc(gy, dt)
    y.backward(gyp)
    close(to_nchw(xg.grad), xr.grad, dt, "conv dgrad")
    close(wg.grad, wr.grad, dt, "conv wgrad")
    if bias:
        close(bg.grad, br.grad, dt, "conv bias grad")


WS_CONVS = [  # N,H,W,Co,bias,relu: 64 input channels -> the weight-stationary persistent kernel (conv3x3_ws.hip)
    (4, 32, 64, 64, True, True),        # one channel block, four tiles per workgroup at 8 workgroups: double-buffered halo
    (3, 16, 48, 27, True, False),       # the DCN offset conv's 27 -> 32 padded channels (32-channel workgroups)
    (2, 48, 32, 256, False, True),      # four channel blocks per tile (the head convs)
    (1, 16, 16, 96, False, False),      # ragged last channel block, a single tile: every halo side is the image border
]


@pytest.mark.parametrize("cfg", WS_CONVS)
def test_conv3x3_weight_stationary(cfg, monkeypatch):
    """bf16 3x3/s1 conv with 64 input channels AND its data gradient when Co == 64 (mirrored taps) through conv3x3_ws_kernel,
    forced onto few workgroups so that every workgroup walks several tiles; compared with torch fp32 on bf16-rounded operands
    and, bit for bit, with the tile kernel (same bf16 products, fp32 accumulation in a different order: equal after rounding
    except for rare 1-ulp flips, so the comparison is by tolerance against torch and by a tight bound against the tile kernel)."""
    N, H, W, Co, bias, relu = cfg
    dt = torch.bfloat16
    x = rng.t_normal(3, f"x{cfg}", (N, 64, H, W))
    w = rng.t_normal(3, f"w{cfg}", (Co, 64, 3, 3), 0, (2.0 / 576) ** 0.5)
    b = rng.t_normal(3, f"b{cfg}", (Co,), 0, 0.1) if bias else None
    gy = rng.t_normal(3, f"g{cfg}", (N, Co, H, W))
    xr, wr = rnd(x, dt).requires_grad_(True), rnd(w, dt).requires_grad_(True)
    yr = F.conv2d(xr, wr, b, 1, 1)
    if relu:
        yr = F.relu(yr)
    yr.backward(rnd(gy, dt))

    def run():
        xg = to_nhwc(x, dt).requires_grad_(True)
        wg = w.to(DEV).requires_grad_(True)
        y = ops().conv2d(xg, wg, b.to(DEV) if bias else None, 1, 1, relu)
        gyp = torch.zeros(y.shape, dtype=dt, device=DEV)
        gyp[..., :Co] = to_nhwc(gy, dt)
        y.backward(gyp)
        torch.cuda.synchronize()
        return y.detach().float().cpu(), xg.grad.detach().float().cpu()

    monkeypatch.setenv("CN_CONV_WS_FORCE", "8")
    y_ws, dx_ws = run()
    monkeypatch.delenv("CN_CONV_WS_FORCE")
    y_tile, dx_tile = run()                          # too few tiles for the size rule: the halo-tile kernel
    close(y_ws.permute(0, 3, 1, 2)[:, :Co], yr, dt, "ws conv fwd")
    close(dx_ws.permute(0, 3, 1, 2), xr.grad, dt, "ws conv dgrad")
    if y_ws.shape[-1] != Co:
        assert float(y_ws[..., Co:].abs().max()) == 0.0, "channel padding must stay zero"
    for a_, b_, what in ((y_ws, y_tile, "fwd"), (dx_ws, dx_tile, "dgrad")):
        s_ = float(b_.abs().max())
        assert float((a_ - b_).abs().max()) <= 2.0 ** -7 * s_, f"ws vs tile kernel, {what}"   # one bf16 ulp at the top of the range


S1_CONVS = [  # npix as N,H,W ; Ci ; Co ; bias ; relu ; transposed : 1x1 convs on the streaming kernel (>= 64 K pixels)
    (4, 128, 128, 256, 80, True, False, False),     # the 80-class head's output conv (96 padded rows, three channel blocks)
    (4, 128, 130, 256, 2, True, False, False),      # wh / reg heads (y_ld = 16), pixel count not a multiple of 32
    (16, 64, 64, 128, 128, False, True, False),
    (16, 64, 64, 128, 256, False, False, False),    # eight channel blocks
    (4, 128, 128, 64, 64, False, False, True),      # data gradient of a 1x1 conv (transposed = the other packing of the weights)
    (4, 128, 128, 32, 64, True, True, False),
]


@pytest.mark.parametrize("cfg", S1_CONVS)
def test_conv1x1_streaming(cfg, monkeypatch):
    """bf16 1x1 convs on conv1x1_stream_kernel (weights in LDS, activations straight from global memory into the MFMA) against
    torch fp32 on bf16-rounded operands and against the implicit-GEMM kernel (CN_DISABLE_CONV1X1_STREAM is read once per process,
    so the second opinion comes from the residual-free identity y(x) linear in x: the kernel on 2x equals 2 y(x) exactly in
    bf16 apart from the ReLU-free bias term, and from the torch reference)."""
    N, H, W, Ci, Co, bias, relu, transposed = cfg
    o = ops()
    dt = torch.bfloat16
    x = rng.t_normal(7, f"x{cfg}", (N, H, W, Ci)).to(dt).to(DEV)
    w = rng.t_normal(7, f"w{cfg}", (Ci, Co, 1, 1) if transposed else (Co, Ci, 1, 1), 0, (2.0 / Ci) ** 0.5)
    b = rng.t_normal(7, f"b{cfg}", (Co,), 0, 0.1) if bias else None
    res = rng.t_normal(7, f"r{cfg}", (N, H, W, (Co + 15) // 16 * 16)).to(dt).to(DEV) if Co % 16 == 0 else None
    wp = o.pack_weight(w.to(DEV), 0 if transposed else 1, dt)
    y = o._igemm(x, wp, b.to(DEV) if bias else None, res, Co, 1, 1, 1, 0, transposed, relu, H, W)
    torch.cuda.synchronize()
    wm = w.to(dt).float()[:, :, 0, 0]
    ref = x.float().cpu().reshape(-1, Ci) @ (wm if transposed else wm.t())
    if bias:
        ref = ref + b
    if res is not None:
        ref = ref + res.float().cpu().reshape(-1, res.shape[-1])[:, :Co]
    if relu:
        ref = ref.clamp_min(0)
    got = y.float().cpu().reshape(-1, y.shape[-1])
    close(got[:, :Co], ref, dt, "1x1 stream")
    if got.shape[1] != Co:
        assert float(got[:, Co:].abs().max()) == 0.0, "channel padding must stay zero"


def test_dma_fed_conv_kernels_repeatable_under_contention():
    """Race screen (tools/attic/ws_stress.py, short form): conv3x3_ws_kernel reads halo tiles that another part of the workgroup
    wrote by LDS-DMA two tiles earlier; a read that races its data would show up as a rare differing tile.  The same full-size
    launch, 40 times, next to 512 MB copies on a second stream, must be bit-identical every time."""
    o = ops()
    dt = torch.bfloat16
    x = rng.t_normal(9, "race_x", (16, 128, 128, 64)).to(dt).to(DEV)
    w = rng.t_normal(9, "race_w", (256, 64, 3, 3), 0, 0.06).to(DEV)
    wp = o.pack_weight(w, 1, dt)
    a = torch.randn(64 * 1024 * 1024, device=DEV)
    b = torch.empty_like(a)
    side = torch.cuda.Stream()
    ref = o._igemm(x, wp, None, None, 256, 3, 3, 1, 1, False, True, 128, 128)
    torch.cuda.synchronize()
    for i in range(40):
        if i % 2 == 0:
            with torch.cuda.stream(side):
                b.copy_(a)
        y = o._igemm(x, wp, None, None, 256, 3, 3, 1, 1, False, True, 128, 128)
        assert torch.equal(y, ref), f"launch {i} differs"
    torch.cuda.synchronize()


def test_conv1x1_streaming_relu_mask():
    """ReLU-backward mask epilogue (relu mode 2: y = res > 0 ? y : 0) of the streaming 1x1 kernel, as the data gradient of a 1x1
    conv behind a deferred ReLU uses it."""
    o = ops()
    dt = torch.bfloat16
    N, H, W, Ci, Co = 4, 128, 128, 64, 128
    dy = rng.t_normal(8, "s1m_dy", (N, H, W, Co)).to(dt).to(DEV)
    xact = rng.t_normal(8, "s1m_x", (N, H, W, Ci)).to(dt).to(DEV)
    w = rng.t_normal(8, "s1m_w", (Co, Ci, 1, 1), 0, 0.1)
    wpd = o.pack_weight(w.to(DEV), 0, dt)                       # rows = Ci, k = Co: the data-gradient packing
    dx = o._igemm(dy, wpd, None, xact, Ci, 1, 1, 1, 0, True, 2, H, W)
    torch.cuda.synchronize()
    ref = dy.float().cpu().reshape(-1, Co) @ w.to(dt).float()[:, :, 0, 0]
    ref = torch.where(xact.float().cpu().reshape(-1, Ci) > 0, ref, torch.zeros_like(ref))
    close(dx.float().cpu().reshape(-1, Ci), ref, dt, "1x1 stream relu mask")


@pytest.mark.parametrize("cfg", [(4, 32, 64), (2, 48, 32), (9, 16, 16)])
def test_two_channel_head_in_one_launch(cfg, monkeypatch):
    """No-grad `HeadConv` with two output channels (width_height / regression; heads.py:9-15) through cn_head2_fwd — the weight-
    stationary 3x3 kernel with the ReLU and the 1x1 conv in its epilogue, 8 partial sums per pixel met by fp32 atomics, no hidden
    tensor — against the two-launch path (CN_DISABLE_HEAD2) and against torch fp32 on bf16-rounded operands.  The fused path does NOT
    round the hidden activation to bf16, so it is the closer of the two to torch; three runs agree to the summation order."""
    from centernet_amd.models.heads import HeadConv
    import centernet_amd.models.heads as heads_mod
    N, H, W = cfg
    dt = torch.bfloat16
    head = HeadConv(2, 64, 256).to(DEV)
    with torch.no_grad():
        head.fc[0].weight.copy_(rnd(rng.t_normal(21, f"w1{cfg}", (256, 64, 3, 3), 0, (2.0 / 576) ** 0.5), dt))
        head.fc[0].bias.copy_(rng.t_normal(21, "b1", (256,), 0, 0.1))
        head.fc[2].weight.copy_(rng.t_normal(21, f"w2{cfg}", (2, 256, 1, 1), 0, 0.05))
        head.fc[2].bias.copy_(rng.t_normal(21, "b2", (2,), 0, 0.5))
    x = rng.t_normal(21, f"x{cfg}", (N, 64, H, W))
    xr = rnd(x, dt)
    ref = F.conv2d(F.relu(F.conv2d(xr, head.fc[0].weight.float().cpu(), head.fc[0].bias.cpu(), 1, 1)), head.fc[2].weight.cpu(), head.fc[2].bias.cpu())
    xg = ops().mark_nhwc(to_nhwc(x, dt), 64)
    monkeypatch.setenv("CN_CONV_WS_FORCE", "8")
    with torch.no_grad():
        fused = [head(xg).float().cpu() for _ in range(3)]
        monkeypatch.setattr(heads_mod, "_NO_HEAD2", True)
        two = head(xg).float().cpu()
    assert fused[0].shape == (N, 2, H, W) and two.shape == (N, 2, H, W)
    sc = float(ref.abs().max())
    assert float((fused[0] - ref).abs().max()) <= 4e-3 * sc, "fused head vs torch"
    assert float((two - ref).abs().max()) <= 1.5e-2 * sc, "two-launch head vs torch (hidden rounded to bf16)"
    assert float((fused[0] - ref).abs().max()) <= float((two - ref).abs().max()) + 1e-6
    for f2 in fused[1:]:
        assert float((f2 - fused[0]).abs().max()) <= 1e-5 * sc, "runs differ only by the order of the eight fp32 partial sums"


def test_hooks_are_per_call_across_host_threads():
    """round-4 VERDICT #8: the ABI keeps no armed state, so host threads can interleave calls with and without hooks freely.  Two
    threads, each on its own stream, launch 200 convolutions in lock-step-free alternation — thread A always WITH a statistics sink,
    thread B always WITHOUT: with the round-4 thread-local handshake this was already separated per thread, but a hooks struct that
    leaked between calls (or a process-wide setting) would show up as B's outputs filling A's sink or as A's `bn_taken` flipping.
    Every A call must report taken with exactly its own sums; no B call may touch a sink."""
    import threading
    o = ops()
    dt = torch.bfloat16
    code = o.dtype_code(dt)
    torch.manual_seed(5)
    xa = torch.randn(2, 24, 40, 64, device=DEV).to(dt)
    xb = torch.randn(2, 24, 40, 64, device=DEV).to(dt)
    w = (torch.randn(64, 64, 3, 3, device=DEV) * 0.05)
    wp = o.pack_weight(w, 1, dt)
    slots = 32
    errs = []

    def conv(x, y, hooks):
        o.call("cn_conv2d_fwd", x, wp, None, None, y, 2, 24, 40, 64, 64, 24, 40, 64, 64, 0, 3, 3, 1, 1, 0, 0, code, code, hooks=hooks)

    ya_ref = torch.empty(2, 24, 40, 64, device=DEV, dtype=dt)
    conv(xa, ya_ref, None)
    torch.cuda.synchronize()
    yf = ya_ref.float().reshape(-1, 64)
    ref = torch.stack([yf.double().sum(0), (yf.double() ** 2).sum(0)])

    def worker(with_hooks):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                y = torch.empty(2, 24, 40, 64, device=DEV, dtype=dt)
                for _ in range(200):
                    if with_hooks:
                        sink = torch.zeros(slots, 2, 64, device=DEV)
                        h = o.Hooks().set(bn_part=sink, bn_slots=slots, bn_C=64)
                        conv(xa, y, h)
                        assert h.bn_taken == 1, "the 64-channel 3x3 kernel has the statistics hook"
                        st.synchronize()
                        got = sink.double().sum(0)
                        assert float(((got - ref).abs() / ref.abs().amax(1, keepdim=True)).max()) < 1e-5
                    else:
                        conv(xb, y, None)
                st.synchronize()
        except BaseException as e:      # noqa: BLE001 (re-raised in the main thread)
            errs.append(e)

    ts = [threading.Thread(target=worker, args=(True,)), threading.Thread(target=worker, args=(False,))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_two_channel_head_is_bit_reproducible_under_the_determinism_flag():
    """round-4 ADVICE: cn_head2_fwd sums eight per-wave partials per pixel with fp32 atomics, so its maps differ in the last bits from
    run to run.  Under torch.use_deterministic_algorithms(True) the host mirror takes the two-launch pair (no atomics): bit-identical
    runs, no-grad and HeadFn (training) forward alike."""
    from centernet_amd.models.heads import HeadConv
    dt = torch.bfloat16
    torch.manual_seed(3)
    head = HeadConv(2, 64, 256).to(DEV)
    xg = ops().mark_nhwc(to_nhwc(torch.randn(4, 64, 32, 64), dt), 64)
    torch.use_deterministic_algorithms(True)
    try:
        with torch.no_grad():
            a, b = head(xg).clone(), head(xg).clone()
        c, d = head(xg).detach().clone(), head(xg).detach().clone()
    finally:
        torch.use_deterministic_algorithms(False)
    assert torch.equal(a, b) and torch.equal(c, d)
    with torch.no_grad():
        fused = head(xg)
    assert float((fused - a).abs().max()) <= 1.5e-2 * float(a.abs().max())


def test_c16_conv_keeps_a_nan():
    """round-4 ADVICE: conv3x3_c16r_kernel's 'ReLU as a lower bound' used fmaxf, which returns the non-NaN operand — a NaN accumulator
    was stored as -inf (ReLU off) or 0 (on) and an isnan screen further down missed it.  The bound now keeps it, like torch's relu."""
    o = ops()
    dt = torch.bfloat16
    x = torch.randn(1, 40, 70, 16).to(dt).to(DEV)
    x[0, 20, 30, 5] = float("nan")
    w = (torch.randn(16, 16, 3, 3) * 0.1).to(DEV)
    wp = o.pack_weight(w, 1, dt)
    for relu in (False, True):
        y = o._igemm(x, wp, None, None, 16, 3, 3, 1, 1, False, relu, 40, 70).float()
        torch.cuda.synchronize()
        assert torch.isnan(y[0, 19:22, 29:32]).all(), "every output whose 3x3 window contains the NaN is NaN"
        assert not torch.isinf(y).any(), "a NaN must not come out as -inf"
        # (the kernel's K packing multiplies zero weights with neighbouring rows, so 0 * NaN may mark a few more outputs of the same
        # columns as NaN — louder than the truth, never quieter; everything away from the spot stays finite)
        far = y.clone()
        far[0, 12:29, 22:39] = 0
        assert torch.isfinite(far).all()


@pytest.mark.parametrize("mode", ["residual_add", "residual_add_relu", "relu_mask", "fp32_rows", "mirrored_taps"])
def test_conv3x3_weight_stationary_epilogues(mode, monkeypatch):
    """Every epilogue variant of conv3x3_ws_kernel (residual add, + ReLU, ReLU-backward mask, fp32 output rows as the DCN offset
    conv writes them, mirrored taps of a data gradient) against the halo-tile kernel on the same operands through cn_conv2d_fwd:
    identical bf16 products and fp32 accumulation in a different order -> at most one output ulp apart."""
    o = ops()
    N, H, W, Co = 3, 32, 48, 64 if mode != "fp32_rows" else 27
    dt = torch.bfloat16
    x = rng.t_normal(5, f"x{mode}", (N, H, W, 64)).to(dt).to(DEV)
    w = rng.t_normal(5, f"w{mode}", (Co, 64, 3, 3), 0, (2.0 / 576) ** 0.5).to(DEV)
    b = rng.t_normal(5, f"b{mode}", (Co,), 0, 0.1).to(DEV)
    res = rng.t_normal(5, f"r{mode}", (N, H, W, 64)).to(dt).to(DEV)
    transposed = mode == "mirrored_taps"
    wp = o.pack_weight(w, 0 if transposed else 1, dt)
    args = dict(residual_add=(b, res, 0), residual_add_relu=(None, res, 1), relu_mask=(None, res, 2), fp32_rows=(b, None, 0),
                mirrored_taps=(None, None, 0))[mode]

    def run():
        y = o._igemm(x, wp, args[0], args[1], Co, 3, 3, 1, 1, transposed, args[2], H, W,
                     out_dtype=torch.float32 if mode == "fp32_rows" else None)
        torch.cuda.synchronize()
        return y.float().cpu()

    monkeypatch.setenv("CN_CONV_WS_FORCE", "8")
    y_ws = run()
    monkeypatch.delenv("CN_CONV_WS_FORCE")
    y_tile = run()
    assert float(y_tile.abs().max()) > 0.5
    tol = 1e-5 if mode == "fp32_rows" else 2.0 ** -7
    assert float((y_ws - y_tile).abs().max()) <= tol * float(y_tile.abs().max()), mode
    if mode == "fp32_rows":
        assert float(y_ws[..., Co:].abs().max()) == 0.0
    if mode == "relu_mask":
        assert bool(((y_ws == 0) == (y_tile == 0)).all())


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 8, 8, 64, 32), (1, 5, 7, 512, 256), (2, 16, 16, 256, 256)])
def test_conv_transpose_4x4_s2(cfg, dt):
    N, H, W, Ci, Co = cfg
    x = rng.t_normal(2, f"x{cfg}", (N, Ci, H, W))
    w = rng.t_normal(2, f"w{cfg}", (Ci, Co, 4, 4), 0, (2.0 / (Ci * 4)) ** 0.5)
    xr, wr = rnd(x, dt).requires_grad_(True), rnd(w, dt).requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, None, 2, 1)
    gy = rng.t_normal(2, f"g{cfg}", tuple(yr.shape))
    yr.backward(rnd(gy, dt))
    xg, wg = to_nhwc(x, dt).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = ops().conv_transpose2d(xg, wg, 2, 1)
    close(to_nchw(y), yr, dt, "convT fwd")
    y.backward(to_nhwc(gy, dt))
    close(to_nchw(xg.grad), xr.grad, dt, "convT dgrad")
    close(wg.grad, wr.grad, dt, "convT wgrad")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 32, 32, 16, 1), (2, 37, 41, 64, 2), (1, 64, 96, 16, 1), (2, 40, 36, 128, 2)])
def test_stem_conv(cfg, dt):
    N, H, W, Co, s = cfg
    x = rng.t_normal(3, f"x{cfg}", (N, 3, H, W))
    w = rng.t_normal(3, f"w{cfg}", (Co, 3, 7, 7), 0, 0.1)
    wr = w.clone().requires_grad_(True)
    yr = F.conv2d(x, wr, None, s, 3)
    gy = rng.t_normal(3, f"g{cfg}", tuple(yr.shape))
    yr.backward(rnd(gy, dt))
    wg = w.to(DEV).requires_grad_(True)
    y = ops().StemConvFn.apply(x.to(DEV), wg, s, 3, dt)
    close(to_nchw(y), yr, dt, "stem fwd")
    y.backward(to_nhwc(gy, dt))
    close(wg.grad, wr.grad, torch.float32 if dt == torch.float32 else dt, "stem wgrad")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 16, 16, 16, True, True), (2, 9, 11, 64, False, True), (3, 8, 8, 512, True, False),
                                 (1, 5, 5, 2048, False, False)])
def test_batchnorm_train(cfg, dt):
    from centernet_amd import nn as hnn
    N, H, W, C, use_res, relu = cfg
    x = rng.t_normal(4, f"x{cfg}", (N, C, H, W), 0.3, 1.7)
    res = rng.t_normal(4, f"r{cfg}", (N, C, H, W)) if use_res else None
    bn_ref = torch.nn.BatchNorm2d(C, momentum=0.1)
    bn = hnn.BatchNorm2d(C).to(DEV)
    with torch.no_grad():
        bn_ref.weight.copy_(rng.t_uniform(4, "g", (C,), 0.5, 1.5)); bn_ref.bias.copy_(rng.t_normal(4, "b", (C,), 0, 0.2))
        bn.weight.copy_(bn_ref.weight); bn.bias.copy_(bn_ref.bias)
    xr = rnd(x, dt).requires_grad_(True)
    rr = rnd(res, dt).requires_grad_(True) if use_res else None
    yr = bn_ref(xr)
    if use_res:
        yr = yr + rr
    if relu:
        yr = F.relu(yr)
    gy = rng.t_normal(4, f"gy{cfg}", tuple(yr.shape))
    yr.backward(rnd(gy, dt))
    xg = to_nhwc(x, dt).requires_grad_(True)
    rg = to_nhwc(res, dt).requires_grad_(True) if use_res else None
    bn.train()
    y = bn(xg, rg, relu)
    close(to_nchw(y), yr, dt, "bn fwd")
    close(bn.running_mean, bn_ref.running_mean, torch.float32, "running_mean")
    close(bn.running_var, bn_ref.running_var, torch.float32, "running_var")
    y.backward(to_nhwc(gy, dt))
    # the ReLU mask of the bf16 run is taken from the bf16-rounded output: exclude elements that round across zero
    close(to_nchw(xg.grad), xr.grad, dt, "bn dx")
    close(bn.weight.grad, bn_ref.weight.grad, dt, "bn dgamma")
    close(bn.bias.grad, bn_ref.bias.grad, dt, "bn dbeta")
    if use_res:
        close(to_nchw(rg.grad), rr.grad, dt, "bn dres")
    assert bn.state_dict()["num_batches_tracked"].item() == 1


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 16, 16, 32, 2, 2, 0), (2, 17, 15, 64, 3, 2, 1), (1, 8, 8, 16, 2, 2, 0)])
def test_maxpool(cfg, dt):
    N, H, W, C, k, s, p = cfg
    x = F.relu(rng.t_normal(5, f"x{cfg}", (N, C, H, W)))          # many exact ties at 0, like post-ReLU maps
    x = rnd(x, dt)
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, k, s, p)
    gy = rnd(rng.t_normal(5, f"g{cfg}", tuple(yr.shape)), dt)
    yr.backward(gy)
    xg = to_nhwc(x, dt).requires_grad_(True)
    y = ops().max_pool(xg, k, s, p)
    assert torch.equal(to_nchw(y), yr.detach()), "maxpool fwd must be exact"
    y.backward(to_nhwc(gy, dt))
    close(to_nchw(xg.grad), xr.grad, dt, "maxpool bwd (first-max tie rule)")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 4, 4, 256, True), (1, 7, 5, 384, True), (2, 1, 1, 512, True), (2, 6, 9, 64, False)])
def test_upsample2x_add(cfg, dt):
    """Hourglass merge: up1 + nn.Upsample(scale_factor=2)(low3) (large_hourglass.py:196-204) and its adjoint."""
    N, H, W, C, with_a = cfg
    low = rnd(rng.t_normal(6, f"l{cfg}", (N, C, H, W)), dt)
    a = rnd(rng.t_normal(6, f"a{cfg}", (N, C, 2 * H, 2 * W)), dt) if with_a else None
    lr = low.clone().requires_grad_(True)
    ar = a.clone().requires_grad_(True) if with_a else None
    up = F.interpolate(lr, scale_factor=2, mode="nearest")
    yr = up + ar if with_a else up
    gy = rnd(rng.t_normal(6, f"g{cfg}", tuple(yr.shape)), dt)
    yr.backward(gy)
    lg = to_nhwc(low, dt).requires_grad_(True)
    ag = to_nhwc(a, dt).requires_grad_(True) if with_a else None
    y = ops().upsample2x_add(ag, lg)
    close(to_nchw(y), yr, dt, "upsample2x_add fwd")
    if not with_a:
        assert torch.equal(to_nchw(y), yr.detach()), "plain nearest up-sampling is a copy"
    y.backward(to_nhwc(gy, dt))
    close(to_nchw(lg.grad), lr.grad, dt, "upsample2x_add d(low) = 2x2 sums")
    if with_a:
        assert torch.equal(to_nchw(ag.grad), ar.grad), "d(a) is dy itself"


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 8, 8, 64, 2, False), (1, 6, 5, 64, 4, True), (2, 4, 4, 256, 2, True),
                                 # the row-walking bf16 weight-gradient kernel of the x2 layers (k = 4): odd sizes, every lane layout
                                 # (8 / 16 / 32 / 64 channel-vector lanes per row), several rows per lane
                                 (3, 7, 9, 128, 2, False), (1, 5, 6, 512, 2, False), (5, 40, 24, 64, 2, True), (2, 3, 2, 1024, 2, False)])
def test_depthwise_up(cfg, dt):
    """depthwise bilinear up-conv, optionally with IDAUp's merge add fused into its store (pose_dla_dcn.py:483-488)"""
    N, H, W, C, f, with_res = cfg
    k = 2 * f
    x = rng.t_normal(6, f"x{cfg}", (N, C, H, W))
    w = rng.t_uniform(6, f"w{cfg}", (C, 1, k, k), 0.0, 0.5)
    xr, wr = rnd(x, dt).requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, None, f, f // 2, groups=C)
    res = rnd(rng.t_normal(6, f"r{cfg}", tuple(yr.shape)), dt) if with_res else None
    rr = res.clone().requires_grad_(True) if with_res else None
    if with_res:
        yr = yr + rr
    gy = rng.t_normal(6, f"g{cfg}", tuple(yr.shape))
    yr.backward(rnd(gy, dt))
    xg, wg = to_nhwc(x, dt).requires_grad_(True), w.to(DEV).requires_grad_(True)
    rg = to_nhwc(res, dt).requires_grad_(True) if with_res else None
    y = ops().DwDeconvFn.apply(xg, wg, f, f // 2, rg)
    close(to_nchw(y), yr, dt, "dwdeconv fwd")
    y.backward(to_nhwc(gy, dt))
    close(to_nchw(xg.grad), xr.grad, dt, "dwdeconv dx")
    close(wg.grad, wr.grad, dt, "dwdeconv dw")
    if with_res:
        assert torch.equal(to_nchw(rg.grad), rr.grad), "d(residual) is dy itself"


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 12, 12, 64, 64), (1, 9, 7, 128, 64), (1, 6, 6, 512, 256), (2, 16, 16, 32, 16),
                                 (1, 40, 36, 64, 64, 6.0), (2, 19, 33, 128, 128, 4.0), (1, 20, 20, 256, 64),
                                 # maps of >= 32x32 pixels with Co = 64 take the 16x16-tile forward (dcn_b2.hip): ragged tiles, displacements
                                 # beyond its 3-pixel window margin (the per-lane far path), several 64-channel blocks of x
                                 (1, 35, 50, 64, 64), (1, 33, 34, 128, 64, 4.0), (2, 64, 48, 256, 64, 1.5),
                                 # ... and a small map keeps the 8x16-tile kernel: far samples in its SECOND 64-channel block (round-5 ADVICE)
                                 (1, 20, 20, 128, 64, 4.0)])
def test_dcnv2(cfg, dt):
    """vs oracle/dcn_ref.py (pure torch); offsets are O(1) so every bilinear corner / border case is exercised."""
    from centernet_amd import nn as hnn
    from oracle.dcn_ref import DCN as RefDCN
    N, H, W, Ci, Co = cfg[:5]
    off_scale = cfg[5] if len(cfg) > 5 else 0.3     # > 3 px displacements exercise the far (global-atomic) path of col2im
    ref = RefDCN(Ci, Co)
    with torch.no_grad():
        ref.weight.copy_(rnd(rng.t_normal(7, f"w{cfg}", (Co, Ci, 3, 3), 0, (2.0 / (Ci * 9)) ** 0.5), dt))
        ref.bias.copy_(rng.t_normal(7, "b", (Co,), 0, 0.1))
        ref.conv_offset_mask.weight.copy_(rnd(rng.t_normal(7, f"ow{cfg}", (27, Ci, 3, 3), 0, 0.6 / (Ci * 9) ** 0.5), dt))
        ref.conv_offset_mask.bias.copy_(rng.t_normal(7, "ob", (27,), 0, off_scale))
    mod = hnn.DCN(Ci, Co).to(DEV)
    mod.load_state_dict(ref.state_dict())
    x = rng.t_normal(7, f"x{cfg}", (N, Ci, H, W))
    xr = rnd(x, dt).requires_grad_(True)
    yr = ref(xr)
    gy = rng.t_normal(7, f"g{cfg}", tuple(yr.shape))
    yr.backward(rnd(gy, dt))
    xg = to_nhwc(x, dt).requires_grad_(True)
    y = mod(xg)
    # bf16: the offsets themselves are rounded to bf16 in HBM, which moves sampling points by up to 2^-8 px
    tol_dt = dt
    close(to_nchw(y), yr, tol_dt, "dcn fwd", scale=float(yr.abs().max()) * (1 if dt == torch.float32 else 3))
    y.backward(to_nhwc(gy, dt))
    sc = (1 if dt == torch.float32 else 4)
    close(to_nchw(xg.grad), xr.grad, dt, "dcn dx", scale=float(xr.grad.abs().max()) * sc)
    close(mod.weight.grad, ref.weight.grad, dt, "dcn dw", scale=float(ref.weight.grad.abs().max()) * sc)
    close(mod.bias.grad, ref.bias.grad, dt, "dcn db", scale=float(ref.bias.grad.abs().max()) * sc)
    close(mod.conv_offset_mask.weight.grad, ref.conv_offset_mask.weight.grad, dt, "dcn d(offset conv w)",
          scale=float(ref.conv_offset_mask.weight.grad.abs().max()) * sc)
    close(mod.conv_offset_mask.bias.grad, ref.conv_offset_mask.bias.grad, dt, "dcn d(offset conv b)",
          scale=float(ref.conv_offset_mask.bias.grad.abs().max()) * sc)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 9, 11, (64, 64), 64), (1, 8, 8, (128, 128, 64, 128), 128), (2, 5, 7, (512, 512, 256), 512),
                                 (1, 6, 6, (16, 48, 32), 40), (2, 16, 16, (256, 256, 128, 256, 16, 32), 256)])
def test_conv1x1_over_concatenation_without_cat(cfg, dt):
    """DLA Root (pose_dla_dcn.py:180-188): conv1x1(torch.cat(children, 1)) with the K loop walking the children — forward, each
    child's data gradient and the column blocks of the weight gradient against torch's cat + conv2d."""
    N, H, W, chans, Co = cfg
    xs = [rng.t_normal(8, f"x{i}{cfg}", (N, c, H, W)) for i, c in enumerate(chans)]
    w = rng.t_normal(8, f"w{cfg}", (Co, sum(chans), 1, 1), 0, (2.0 / sum(chans)) ** 0.5)
    xr = [rnd(x, dt).requires_grad_(True) for x in xs]
    wr = rnd(w, dt).requires_grad_(True)
    yr = F.conv2d(torch.cat(xr, 1), wr)
    gy = rng.t_normal(8, f"g{cfg}", tuple(yr.shape))
    yr.backward(rnd(gy, dt))
    xg = [to_nhwc(x, dt).requires_grad_(True) for x in xs]
    wg = w.to(DEV).requires_grad_(True)
    y = ops().conv1x1_cat(xg, wg)
    assert y.shape[-1] == (Co + 15) // 16 * 16 and float(y[..., Co:].abs().max() if y.shape[-1] > Co else 0) == 0.0
    close(to_nchw(y[..., :Co]), yr, dt, "cat conv fwd")
    gyp = torch.zeros(N, y.shape[-1], H, W); gyp[:, :Co] = gy
    y.backward(to_nhwc(gyp, dt))
    for i, (a, b) in enumerate(zip(xg, xr)):
        close(to_nchw(a.grad), b.grad, dt, f"cat conv dx[{i}]")
    close(wg.grad, wr.grad, dt, "cat conv dw")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("chans", [(16, 16), (64, 64), (128, 128)])      # gather kernel, 64-channel tile path, LDS-resident tile kernel
def test_dcn_border_known_answers_through_the_c_abi(chans, dt):
    """DCNv2 is parity-unpinned (its source is not under /root/reference), so the HIP forward is held to the same HAND-computed
    border vectors as the oracle (tests/conftest.py:dcn_border_vectors): fractional sample points at all four borders / corners
    and py = -1, py = H, px = -1, px = W exactly, straight through cn_dcn_fwd."""
    from conftest import dcn_border_vectors
    from centernet_amd._hip import call, dtype_code
    x, off, mask, weight, want = dcn_border_vectors()
    Ci, Co = chans
    N, H, W = 1, 4, 4
    xg = torch.zeros(N, H, W, Ci, dtype=dt, device=DEV)
    xg[..., 0] = x[0, 0].to(dt).to(DEV)                       # the image is channel 0; the other channels are zero
    om = torch.zeros(N, H, W, 32, dtype=torch.float32, device=DEV)
    om[..., :18] = off[0].permute(1, 2, 0).float().to(DEV)
    om[..., 18:27] = 30.0                                     # mask logits: sigmoid(30) == 1.0f
    wfull = torch.zeros(Co, Ci, 3, 3)
    wfull[0, 0] = weight[0, 0].float()
    wfull[1, 0, 1, 1] = -2.0                                  # a second output channel: -2 x the same sample
    wp = ops().pack_weight(wfull.to(DEV), 1, dt)
    bias = torch.zeros(Co, dtype=torch.float32, device=DEV)
    bias[1] = 0.5
    y = torch.zeros(N, H, W, Co, dtype=dt, device=DEV)
    call("cn_dcn_fwd", xg, om, wp, bias, y, N, H, W, Ci, Ci, Co, Co, 32, 0, dtype_code(dt))
    got = y.float().cpu()
    tol = 1e-6 if dt == torch.float32 else 2.0 ** -7 * 8.5
    assert float((got[0, :, :, 0].double() - want).abs().max()) <= tol, (got[0, :, :, 0], want)
    assert float((got[0, :, :, 1].double() - (0.5 - 2.0 * want)).abs().max()) <= 2 * tol + (0 if dt == torch.float32 else 0.07)
    assert float(got[0, :, :, 2:].abs().max()) == 0.0
    # mask logit 0 -> sigmoid 0.5 halves every sample
    om[..., 18:27] = 0.0
    call("cn_dcn_fwd", xg, om, wp, bias, y, N, H, W, Ci, Ci, Co, Co, 32, 0, dtype_code(dt))
    assert float((y.float().cpu()[0, :, :, 0].double() - 0.5 * want).abs().max()) <= tol


def test_dcn_fwd_tile_kernel_matches_gather_kernel():
    """The LDS-resident forward (dcn_fwd_tile.hip) and the global-gather forward (dcn_fused.hip) on the same inputs: they may
    differ by one bf16 ulp on a few outputs (fma order of the 4-corner blend).  Separate processes: the switch is read once."""
    import os, subprocess, sys, tempfile
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for cfg in (["3", "33", "45", "64", "64", "0.4"], ["2", "19", "37", "128", "128", "3.0"], ["2", "16", "16", "256", "64", "1.0"]):
        outs = []
        for env in ({"CN_FORCE_DCN_FWD_TILE": "1"}, {"CN_DISABLE_DCN_FWD_TILE": "1"}):
            f = tempfile.NamedTemporaryFile(suffix=".pt", delete=False).name
            r = subprocess.run([sys.executable, os.path.join(repo, "tools", "dcn_fwd_ab.py"), *cfg, f], env=dict(os.environ, **env),
                               capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append(torch.load(f))
            os.unlink(f)
        a, b = outs
        d = (a - b).abs()
        assert float(b.abs().max()) > 1.0 and float(d.max()) <= 2 ** -6 * float(b.abs().max()), cfg     # <= ~1 bf16 ulp at the top of the range
        assert float((d > 0).float().mean()) < 2e-3, cfg


def test_dcn_fwd_b2_kernel_matches_blend_matrix_kernel():
    """The 16x16-tile forward (dcn_b2.hip: window fragments from the LDS halo, no geometry table, four waves per SIMD) against the
    8x16-tile blend-matrix forward (dcn_bm.hip) on the same inputs: both blend with bf16 weights on the matrix cores, so they may
    differ by one bf16 ulp on a few outputs (fp32 accumulation order; the far path's exact fp32 weights where only ONE of them takes
    it: the window margins differ, 3 vs 4 pixels).  Separate processes: the switch is read once."""
    import os, subprocess, sys, tempfile
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for cfg in (["2", "48", "80", "64", "64", "0.4"], ["1", "37", "53", "64", "64", "2.5"], ["2", "32", "32", "128", "64", "1.0"],
                ["1", "64", "64", "256", "64", "0.2"]):
        outs = []
        for env in ({}, {"CN_DISABLE_DCN_FWD_B2": "1"}):
            f = tempfile.NamedTemporaryFile(suffix=".pt", delete=False).name
            r = subprocess.run([sys.executable, os.path.join(repo, "tools", "dcn_fwd_ab.py"), *cfg, f], env=dict(os.environ, **env),
                               capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append(torch.load(f))
            os.unlink(f)
        a, b = outs
        d = (a - b).abs()
        assert float(b.abs().max()) > 1.0 and float(d.max()) <= 2 ** -6 * float(b.abs().max()), cfg     # <= ~1 bf16 ulp at the top of the range
        # offsets of several pixels: many samples take the exact-weight far path in ONE of the two kernels only (window margin 3 vs 4)
        assert float((d > 0).float().mean()) < (5e-3 if float(cfg[5]) < 1.0 else 0.12), cfg


def test_dcn_zero_init_is_half_conv():
    """Known-answer test 1 of SURVEY Appendix A on the HIP kernel (the state of every DCN at the start of training)."""
    from centernet_amd import nn as hnn
    mod = hnn.DCN(32, 48).to(DEV)
    x = rng.t_normal(8, "x", (2, 32, 10, 10))
    y = mod(to_nhwc(x, torch.float32))
    ref = 0.5 * F.conv2d(x, mod.weight.detach().cpu(), None, 1, 1) + mod.bias.detach().cpu().view(1, -1, 1, 1)
    close(to_nchw(y), ref, torch.float32, "dcn zero-init")


@pytest.mark.parametrize("dt", DTYPES)
def test_glue_ops(dt):
    o = ops()
    a = rng.t_normal(9, "a", (2, 32, 7, 9)); b = rng.t_normal(9, "b", (2, 16, 7, 9)); c = rng.t_normal(9, "c", (2, 64, 7, 9))
    ag, bg, cg = (to_nhwc(t, dt).requires_grad_(True) for t in (a, b, c))
    cat = o.concat([ag, bg, cg])
    assert torch.equal(to_nchw(cat), torch.cat([rnd(a, dt), rnd(b, dt), rnd(c, dt)], 1))
    g = rng.t_normal(9, "g", (2, 112, 7, 9))
    cat.backward(to_nhwc(g, dt))
    assert torch.equal(to_nchw(bg.grad), rnd(g, dt)[:, 32:48])
    s = o.add(ag.detach(), to_nhwc(rng.t_normal(9, "a2", (2, 32, 7, 9)), dt))
    close(to_nchw(s), rnd(a, dt) + rnd(rng.t_normal(9, "a2", (2, 32, 7, 9)), dt), dt, "add")
    cg = to_nhwc(c, dt).requires_grad_(True)
    y = o.ToNCHWFn.apply(cg, 50)
    assert y.dtype == torch.float32 and torch.equal(y.cpu(), rnd(c, dt)[:, :50])
    y.backward(g[:, :50].to(DEV).contiguous())
    gg = to_nchw(cg.grad)
    assert torch.equal(gg[:, :50], rnd(g[:, :50], dt)) and float(gg[:, 50:].abs().max()) == 0


def test_adam_matches_torch():
    from centernet_amd.engine import FlatAdam
    ps = [torch.nn.Parameter(rng.t_normal(10, f"p{i}", s).to(DEV)) for i, s in enumerate([(33, 7), (5,), (64, 3, 3, 3)])]
    qs = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    opt, ref = FlatAdam(ps, lr=1e-2), torch.optim.Adam(qs, lr=1e-2)
    for it in range(3):
        opt.zero_grad(); ref.zero_grad()
        for i, (p, q) in enumerate(zip(ps, qs)):
            g = rng.t_normal(11 + it, f"g{i}", tuple(q.shape))
            p.grad.add_(g.to(DEV)); q.grad = g.clone()
        opt.step(); ref.step()
    for p, q in zip(ps, qs):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("sizes", [(64, 32, 5), (64, 256, 2), (64, 256, 80), (32, 256, 17), (32, 256, 34)])
def test_head_conv_fused_relu_backward(dt, sizes):
    """heads.py:4-25: conv3x3+ReLU -> conv1x1; the hidden ReLU's backward runs as the mask mode (relu=2) of the 1x1
    conv's data-gradient epilogue (GEMM kernel; 256-wide hidden layers in bf16: the VALU stream for 1-/2-channel heads, the MFMA
    stream for 17 .. 96 channels).  Reference: plain torch autograd on the same (rounded) operands."""
    from centernet_amd.models.heads import HeadConv
    torch.manual_seed(0)
    N, H, W = 2, 9, 11
    Cin, Cmid, Cout = sizes
    head = HeadConv(Cout, Cin, Cmid).to(DEV)
    x = rng.t_normal(21, "hx", (N, Cin, H, W))
    with torch.no_grad():
        for i, prm in enumerate(head.parameters()):
            prm.copy_(rng.t_normal(21, f"hp{i}", tuple(prm.shape), 0, 0.1))
    w1, b1, w2, b2 = [p.detach().cpu().clone().requires_grad_(True) for p in head.parameters()]
    xr = rnd(x, dt).requires_grad_(True)
    hr = F.relu(F.conv2d(xr, rnd(w1, dt), b1, 1, 1))
    yr = F.conv2d(rnd(hr, dt) + (hr - hr.detach()), rnd(w2, dt), b2)          # hidden rounded like the kernel's output
    gy = rng.t_normal(21, "hg", tuple(yr.shape))
    yr.backward(gy)
    xg = to_nhwc(x, dt).requires_grad_(True)
    y = head(xg)                                                              # NCHW fp32 [N, Cout, H, W]
    close(y.detach().cpu(), yr, dt, "head fwd")
    y.backward(gy.to(DEV))
    close(to_nchw(xg.grad), xr.grad, dt, "head dgrad (mask mode)")


def test_pack_weight_batch_matches_single():
    """ops.PackArena / cn_pack_weight_batch: one launch, same bits as per-layer cn_pack_weight in all three modes."""
    o = ops()
    ws = [rng.t_normal(31, f"w{i}", shp).to(DEV) for i, shp in enumerate([(64, 64, 3, 3), (27, 64, 3, 3), (80, 256, 1, 1), (16, 3, 7, 7)])]
    arena = o.PackArena()
    o.PackArena.current, arena.recording = arena, True
    singles = []
    for w in ws:
        for mode in (0, 1, 2):
            singles.append((w, mode, o.pack_weight(w, mode, torch.bfloat16)))
    arena.build()
    for v in arena.slots.values():                # poison the destinations, then repack in one launch
        v.fill_(7.0)
    arena.repack()
    for w, mode, ref in singles:
        got = o.pack_weight(w, mode, torch.bfloat16)
        assert got.data_ptr() != ref.data_ptr() and torch.equal(got, ref), f"mode {mode} shape {tuple(w.shape)}"
    o.PackArena.current = None


def test_wgrad_thin_grids_same_result():
    """cn_hooks.wgrad_blocks only reshapes the split-K grid of THAT call: the gradients must not change (fp32 atomics: tolerance),
    and a call without hooks right after a thin one runs the default grid again (nothing is remembered in the library)."""
    o = ops()
    x = to_nhwc(rng.t_normal(32, "x", (2, 64, 24, 40)), torch.bfloat16)
    dy = to_nhwc(rng.t_normal(32, "g", (2, 64, 24, 40)), torch.bfloat16)
    outs = []
    code = o.dtype_code(torch.bfloat16)
    for blocks in (1536, 48, None):
        dwp, db = torch.zeros(64, 9 * 64, device=DEV), torch.zeros(64, device=DEV)
        hooks = None if blocks is None else o.Hooks().set(wgrad_blocks=blocks)
        o.call("cn_conv2d_wgrad", x, dy, dwp, db, 2, 24, 40, 64, 64, 24, 40, 64, 64, 3, 3, 1, 1, code, hooks=hooks)
        outs.append((dwp, db))
    close(outs[1][0], outs[0][0], torch.float32, "thin-grid dW")
    close(outs[1][1], outs[0][1], torch.float32, "thin-grid db")
    torch.cuda.synchronize()
    assert torch.equal(outs[2][0], outs[0][0]) or float((outs[2][0] - outs[0][0]).abs().max()) < 1e-3 * float(outs[0][0].abs().max())
    # the slab form: the scratch size follows the grid given to the size query, and a launch with a larger grid than its scratch
    # was sized for is refused (CN_EWORKSPACE), never overrun
    dims = (2, 24, 40, 64, 64, 24, 40, 64, 64, 3, 3, 1, 1, code)
    n_thin, n_wide = (int(o._hip.query("cn_conv2d_wgrad_direct_bytes_h", *dims, b)) for b in (8, 1536))
    assert 0 < n_thin <= n_wide and n_wide == int(o._hip.query("cn_conv2d_wgrad_direct_bytes", *dims))
    ws = torch.empty(n_wide, dtype=torch.uint8, device=DEV)
    res = []
    for b, n in ((8, n_thin), (1536, n_wide)):
        dw = torch.zeros(64, 64, 3, 3, device=DEV)
        o.call("cn_conv2d_wgrad_direct", x, dy, dw, None, 0, ws, n, *dims, hooks=o.Hooks().set(wgrad_blocks=b))
        res.append(dw)
    close(res[0], res[1], torch.float32, "slab-form dW at two grids")
    close(res[1], o.unpack_wgrad(outs[0][0], 64, 64, 3, 3), torch.float32, "slab form vs packed form")
    if n_thin < n_wide:
        with pytest.raises(RuntimeError, match="workspace"):
            o.call("cn_conv2d_wgrad_direct", x, dy, torch.zeros(64, 64, 3, 3, device=DEV), None, 0, ws, n_thin, *dims, hooks=o.Hooks().set(wgrad_blocks=1536))


def test_dcn_far_buffer_is_left_clean():
    """lazy dx_far protocol: after a backward with samples displaced > 3 px the persistent scratch is all zeros again."""
    import os
    from centernet_amd import nn as hnn
    if os.environ.get("CN_DCN_UNFUSED"):
        pytest.skip("the A/B column-tensor pipeline clears its own dx_far")
    o = ops()
    m = hnn.DCN(64, 64).to(DEV)
    with torch.no_grad():
        m.conv_offset_mask.weight.normal_(0, 0.05)
        m.conv_offset_mask.bias[:18].fill_(6.0)        # every sample lands ~6 px away: the far path
    x = to_nhwc(rng.t_normal(33, "x", (1, 64, 20, 24)), torch.bfloat16).requires_grad_(True)
    y = m(x)
    y.backward(torch.ones_like(y))
    buf = o._FAR_BUFFERS[((1, 20, 24, 64), str(x.device))]
    assert float(buf.abs().max()) == 0.0
    assert float(x.grad.float().abs().sum()) > 0.0


BN_STAT_PRODUCERS = [  # kind, N, H, W, Ci, Co, k, stride  (which kernel: see the comment)
    ("conv", 2, 16, 16, 128, 128, 3, 1),      # halo-tile 3x3 kernel (conv3x3s1_kernel), LDS-staged epilogue
    ("conv", 2, 13, 11, 256, 256, 3, 1),      # ragged tiles
    ("conv", 2, 16, 16, 64, 128, 3, 2),       # implicit GEMM (stride 2)
    ("conv", 1, 12, 20, 32, 64, 1, 1),        # implicit GEMM 1x1
    ("cat", 2, 8, 8, (128, 128), 128, 1, 1),  # DLA Root: conv1x1 over a concatenation
    ("dcn", 2, 12, 20, 64, 64, 3, 1),         # blend-matrix DCNv2 forward
    ("dcn", 1, 9, 7, 64, 32, 3, 1),
    ("dcn", 2, 12, 20, 128, 128, 3, 1),       # LDS-resident tile DCNv2 forward (>= 128 channels on both sides)
    ("dcn", 1, 9, 7, 256, 128, 3, 1),
    ("dcn", 2, 12, 20, 128, 64, 3, 1),        # blend-matrix forward over two 64-channel blocks of x (dcn_fwd_bm_kernel<2, true>)
    ("dcn", 2, 12, 20, 192, 64, 3, 1),        # gather DCNv2 forward with the LDS-staged epilogue (192 -> 64: neither matrix-core-blend nor tile kernel)
    ("dcn", 1, 40, 36, 64, 64, 3, 1),         # 16x16-tile forward (dcn_fwd_b2_kernel), ragged tiles
    ("dcn", 1, 33, 34, 128, 64, 3, 1),        # ... over two 64-channel blocks of x
    ("conv", 2, 9, 70, 16, 16, 3, 1),         # row-walking 16-channel kernel (DLA level0)
    ("conv", 2, 38, 70, 16, 32, 3, 2),        # ... stride 2, two output-channel blocks (DLA level1)
    ("conv", 1, 64, 200, 16, 32, 3, 2),       # ... with interior waves
    ("conv", 1, 52, 130, 16, 16, 3, 1),
    ("stem", 2, 37, 41, 3, 16, 7, 1),         # 7x7 stem on the NCHW fp32 image (DLA base_layer)
]


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("cfg", BN_STAT_PRODUCERS)
def test_bn_statistics_from_the_producer_epilogue(cfg, fused, monkeypatch):
    """N1 (north_star: conv + BN + ReLU fused, training mode): the producing kernel fills the statistics sink with sum / sum of
    squares of the bf16 values it stores; the BN on top then equals cn_bn_train_fwd on the same tensor.  fused: ONE launch
    (cn_bn_train_fwd_sink reduces the sink in the apply kernel's prologue; the sink is zeroed by the NEXT sink-consuming launch —
    here the BN backward, cn_bn_train_bwd_sink); not fused: finalize + apply (cn_bn_train_fwd_stats hands the sink back all-zero).
    Reference: msra_resnet.py:29-58 / pose_dla_dcn.py:55-68, 435-454 (conv -> nn.BatchNorm2d in training)."""
    from centernet_amd import nn as hnn
    o = ops()
    monkeypatch.setattr(o.BnStats, "fused", fused)
    kind, N, H, W, Ci, Co, k, stride = cfg
    dt = torch.bfloat16
    torch.manual_seed(3)
    if kind == "cat":
        xs = [to_nhwc(torch.randn(N, c, H, W), dt).requires_grad_(True) for c in Ci]
        conv = hnn.Conv2d(sum(Ci), Co, 1).to(DEV)
        produce = lambda flag: o.conv1x1_cat(xs, conv.weight, bn_stats=flag)
    elif kind == "stem":
        img = torch.randn(N, Ci, H, W, device=DEV)
        stem = hnn.StemConv(Ci, Co, k, stride, k // 2).to(DEV)
        produce = lambda flag: stem(img, dt, bn_stats=flag)
    elif kind == "dcn":
        x = to_nhwc(torch.randn(N, Ci, H, W), dt).requires_grad_(True)
        dcn = hnn.DCN(Ci, Co).to(DEV)
        torch.nn.init.normal_(dcn.conv_offset_mask.weight, std=0.02)
        produce = lambda flag: dcn(x, bn_stats=flag)
    else:
        x = to_nhwc(torch.randn(N, Ci, H, W), dt).requires_grad_(True)
        conv = hnn.Conv2d(Ci, Co, k, stride, k // 2).to(DEV)
        produce = lambda flag: conv(x, bn_stats=flag)
    y0 = produce(False)
    assert getattr(y0, "_bn_part", None) is None
    y1 = produce(True)
    part = getattr(y1, "_bn_part", None)
    assert part is not None, "this producer's kernel is expected to have the statistics hook"
    assert torch.equal(y0.detach(), y1.detach()), "the hook must not change the stored values"
    yf = y1.detach().float().reshape(-1, y1.shape[-1])
    got = part.double().sum(0).cpu()                                        # [2][C]
    ref = torch.stack([yf.double().sum(0), (yf.double() ** 2).sum(0)]).cpu()
    scale = ref.abs().amax(1, keepdim=True).clamp_min(1e-6)
    assert float(((got - ref).abs() / scale).max()) < 1e-5, "sum / sum of squares of the stored values"
    # BN on top: fused statistics vs the stand-alone statistics pass
    bn_a, bn_b = hnn.BatchNorm2d(y1.shape[-1]).to(DEV).train(), hnn.BatchNorm2d(y1.shape[-1]).to(DEV).train()
    za = bn_a(y1, None, True)                                               # consumes the sink
    if not fused:
        assert float(part.abs().max()) == 0.0, "the sink is handed back all-zero"
    zb = bn_b(y0, None, True)
    close(za, zb, dt, "BN(conv) with epilogue statistics vs stand-alone statistics", scale=float(zb.detach().float().abs().max()))
    assert torch.allclose(bn_a.running_mean, bn_b.running_mean, rtol=1e-5, atol=1e-6) and torch.allclose(bn_a.running_var, bn_b.running_var, rtol=1e-4, atol=1e-6)
    # backward: same saved statistics; sums through a sink reduced by the apply pass (fused) vs partial rows + finalize launch
    ga = torch.autograd.grad(za.float().square().sum(), [y1, bn_a.weight, bn_a.bias])
    gb = torch.autograd.grad(zb.float().square().sum(), [y0, bn_b.weight, bn_b.bias])
    assert float(part.abs().max()) == 0.0, "the retired sink is zeroed by the next sink-consuming launch"
    close(ga[0], gb[0], dt, "BN backward dx", scale=float(gb[0].float().abs().max()))
    for u, v, nm in ((ga[1], gb[1], "dgamma"), (ga[2], gb[2], "dbeta")):
        assert float((u - v).abs().max()) <= 2e-2 * float(v.abs().max()) + 1e-3, nm


@pytest.mark.parametrize("cfg", [(2, 40, 70, 16, 1, True), (1, 75, 45, 32, 2, True), (2, 9, 33, 16, 1, False), (1, 64, 200, 32, 2, True)])
def test_conv_applies_previous_bn_on_load(cfg):
    """cn_hooks.pre_ss: the 16-input-channel kernels take the RAW output of the previous conv and apply that layer's BN
    (+ ReLU) on the way into the matrix cores; the result must be BIT-identical to convolving the tensor cn_scale_shift_act stores
    (same fma, same rounding, zero padding after the affine map).  Shapes without the hook must fail loudly."""
    o = ops()
    N, H, W, Co, stride, relu = cfg
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, H, W, 16, generator=g).to(dt).to(DEV)
    ss = torch.cat([torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g) * 0.5]).to(DEV).contiguous()
    w = (torch.randn(Co, 16, 3, 3, generator=g) * 0.1).to(DEV)
    wp = o.pack_weight(w, 1, dt)
    OH, OW = o.conv_out(H, 3, stride, 1), o.conv_out(W, 3, stride, 1)
    act = torch.empty_like(x)
    o.call("cn_scale_shift_act", x, None, act, ss[:16].contiguous(), ss[16:].contiguous(), N * H * W, 16, int(relu), o.dtype_code(dt))
    ref = o._igemm(act, wp, None, None, Co, 3, 3, stride, 1, False, False, OH, OW)
    got = o._igemm(x, wp, None, None, Co, 3, 3, stride, 1, False, False, OH, OW, pre=(ss, relu))
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    x64 = torch.randn(1, 8, 8, 64, generator=g).to(dt).to(DEV)
    w64 = o.pack_weight(torch.randn(64, 64, 3, 3, generator=g).to(DEV), 1, dt)
    ss64 = torch.ones(128, device=DEV)
    with pytest.raises(RuntimeError, match="pre-affine"):
        o._igemm(x64, w64, None, None, 64, 3, 3, 1, 1, False, False, 8, 8, pre=(ss64, True))
    y = o._igemm(x64, w64, None, None, 64, 3, 3, 1, 1, False, False, 8, 8)      # (the refused call left nothing behind)
    assert torch.isfinite(y.float()).all()


@pytest.mark.parametrize("grad", [True, False])
def test_dla_base_chain_leaves_bn_apply_to_the_next_conv(grad, monkeypatch):
    """pose_dla_dcn.py:283-296: base_layer -> level0 -> level1 = conv -> BN -> ReLU x 3 on 16-channel full-resolution tensors.  In
    training mode the two 16-channel BN apply passes are left to the consuming conv (ops.BnDeferFn + cn_hooks.pre_ss): forward
    values, running statistics and every gradient must match the unfused chain (same arithmetic; the batch statistics are reduced by
    a different kernel, so last-bit differences of scale / shift may flip individual bf16 roundings)."""
    from centernet_amd import nn as hnn
    from centernet_amd.models.backbones.pose_dla_dcn import DLA
    o = ops()
    dt = torch.bfloat16
    torch.manual_seed(5)
    net = DLA([1, 1, 1, 2, 2, 1], [16, 32, 64, 128, 256, 512], compute_dtype=dt).to(DEV).train()
    img = torch.randn(2, 3, 72, 104, device=DEV)

    def run(defer, stem_fused=False, bwd_hook=False):
        monkeypatch.setattr(hnn, "BN_DEFER", defer)
        monkeypatch.setattr(hnn, "STEM_BN_FUSED", stem_fused)
        monkeypatch.setattr(o.BnBwdSinks, "enabled", bwd_hook)
        taken = []
        real_note = o.BnBwdSinks.note.__func__
        monkeypatch.setattr(o.BnBwdSinks, "note", classmethod(lambda cls, t, sink: (taken.append(1), real_note(cls, t, sink))[1]))
        for m in net.modules():
            if isinstance(m, hnn.BatchNorm2d):
                m.reset_running_stats()
        net.zero_grad(set_to_none=True)
        calls = []
        real = o.batch_norm_defer
        monkeypatch.setattr(o, "batch_norm_defer", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        with torch.set_grad_enabled(grad):
            x = hnn.stem_bn_act(net.base_layer[0], net.base_layer[1], img, dt, defer=True)
            x = net._run_conv_level(net.level0, x, net.level1[0])
            x = net._run_conv_level(net.level1, x)
        monkeypatch.setattr(o, "batch_norm_defer", real)
        assert getattr(x, "_cn_pre", None) is None, "level1's output is a stored activation"
        assert len(calls) == ((1 if stem_fused else 2) if defer else 0)
        out = {"y": x.detach().float().clone(), "rm0": net.base_layer[1].running_mean.clone(), "rv1": net.level0[1].running_var.clone()}
        if grad:
            g = torch.randn(x.shape, generator=torch.Generator().manual_seed(1)).to(DEV).to(dt)
            x.backward(g)
            assert len(taken) == (2 if (bwd_hook and defer) else 0), "both 16-channel data-gradient kernels have the BN-backward statistics hook"
            assert not o.BnBwdSinks.entries
            for name in ("base_layer.0.weight", "base_layer.1.weight", "base_layer.1.bias", "level0.0.weight", "level0.1.weight",
                         "level0.1.bias", "level1.0.weight", "level1.1.weight"):
                out[name] = net.get_parameter(name).grad.detach().float().clone()
        return out

    b = run(False)
    # second: the stem's BN entirely inside its neighbours (ops.StemBnDeferFn + cn_stem_conv_wgrad_bn); third / fourth: the BN backward
    # statistics of both 16-channel BNs from the epilogue of the kernel that produces their gradient (cn_hooks.bnb_part)
    for a in (run(True), run(True, True), run(True, False, True), run(True, True, True)):
        for k in b:
            scale = float(b[k].abs().max())
            err = float((a[k] - b[k]).abs().max()) / max(scale, 1e-6)
            assert err < (1e-5 if k.startswith("r") else 2e-2), f"{k}: {err:.3e}"


def test_bn_statistics_hook_can_be_declined(monkeypatch):
    """kernels without the hook (weight-stationary 3x3, fp32 compute) report `not taken`: BN then reads x itself"""
    from centernet_amd import nn as hnn
    x = to_nhwc(torch.randn(1, 16, 8, 8), torch.float32).requires_grad_(True)
    conv = hnn.Conv2d(16, 32, 3, 1, 1).to(DEV)
    y = conv(x, bn_stats=True)
    assert getattr(y, "_bn_part", None) is None
    bn = hnn.BatchNorm2d(32).to(DEV).train()
    z = bn(y, None, True)
    assert bool(torch.isfinite(z).all())


@pytest.mark.parametrize("dt", DTYPES)
def test_shared_tensor_gradient_cells_small_graph(dt):
    """ops.share / ops.GradCell on a graph with one consumer of every kind, against autograd's own accumulation over the same
    operators (each of which is held to its torch reference elsewhere in this file)
    (pose_dla_dcn.py:245-262: a Tree's input goes through `downsample` and a stride-1 block whose skip path ends in bn2's residual
    input, and it is a child of a Root):  x -> { max-pool (cn_maxpool_bwd_acc), conv1 with the skip path of a residual block
    (cn_bn_train_bwd_acc for the skip, the conv's residual slot), 1x1 Root over [block, x] (residual slot), a pass-through use }."""
    from centernet_amd import nn as hnn
    o = ops()
    N, C, H, W = 2, 32, 12, 16
    torch.manual_seed(7)
    x0 = rnd(torch.randn(N, C, H, W), dt)
    conv1, conv2, root = hnn.Conv2d(C, C, 3, 1, 1), hnn.Conv2d(C, C, 3, 1, 1), hnn.Conv2d(2 * C, C, 1)
    bn1, bn2, bnr = hnn.BatchNorm2d(C), hnn.BatchNorm2d(C), hnn.BatchNorm2d(C)
    mods = [conv1, conv2, root, bn1, bn2, bnr]
    for m in mods:
        m.to(DEV).train()
        for p in m.parameters():
            p.data = rnd(p.data.cpu(), dt).to(DEV) if p.dim() > 1 else p.data

    def net(x, shared):
        if shared:
            x = o.share(x)
        pooled = o.max_pool(x, 2, 2)
        y, skip = hnn.conv_bn_act_skip(conv1, bn1, x)
        blk = hnn.conv_bn_act(conv2, bn2, y, skip, True)
        r = hnn.cat_conv_bn_act(root, bnr, [blk, x], None, True)
        return pooled, r, x

    outs = {}
    for shared in (False, True):
        for m in mods:
            m.zero_grad(set_to_none=True)
        old, o.GradCell.adds = o.GradCell.enabled, 0
        o.GradCell.enabled = shared
        try:
            xg = to_nhwc(x0, dt).requires_grad_(True)
            xin = xg * 1                                    # a non-leaf producer, like a layer's output
            pooled, r, xa = net(xin, shared)
            torch.manual_seed(11)
            gp, gr, gx = (torch.randn_like(pooled.float()).to(dt), torch.randn_like(r.float()).to(dt), torch.randn_like(xa.float()).to(dt))
            torch.autograd.backward([pooled, r, xa], [gp, gr, gx])     # `xa` itself is read by a cell-unaware consumer too
            outs[shared] = (xg.grad.detach().float().cpu(), [p.grad.detach().float().cpu() for m in mods for p in m.parameters()], o.GradCell.adds)
        finally:
            o.GradCell.enabled = old
    (g0, p0, a0), (g1, p1, a1) = outs[False], outs[True]
    assert a0 == 0 and a1 <= 1, (a0, a1)                    # the cell-unaware use joins in ShareFn.backward: at most one add
    close(g1, g0, dt, "d x through shared-tensor cells vs autograd accumulation")
    for u, v in zip(p1, p0):
        close(u, v, dt, "parameter gradient", scale=max(1e-6, float(v.abs().max())))


# ---------------------------------------------------------------------------------------------- head backward on gathered rows
def _head_reference(x, w1, b1, w2, b2, ind, mask, target):
    """heads.py:4-25 + RegL1Loss (utils/losses.py:53-63) in plain torch (NCHW fp32, autograd's dense backward)."""
    import torch.nn.functional as F
    h = F.relu(F.conv2d(x, w1, b1, padding=1))
    out = F.conv2d(h, w2, b2)
    B, C = out.shape[:2]
    pred = out.view(B, C, -1).gather(2, ind.unsqueeze(1).expand(B, C, ind.shape[1])).permute(0, 2, 1)
    m = mask.unsqueeze(2).expand_as(pred).float()
    return F.l1_loss(pred * m, target * m, reduction="sum") / (m.sum() + 1e-4)


@pytest.mark.parametrize("dt,C,M,hw", [(torch.float32, 2, 16, 24), (torch.float32, 34, 12, 16), (torch.bfloat16, 2, 32, 32)])
def test_head_backward_on_gathered_rows(dt, C, M, hw):
    """ops.HeadFn under a gather-type loss: the row path (csrc/head_sparse.hip) against torch's dense autograd of the same head —
    input gradient (through a shared-tensor cell next to a dense sibling), both weight and bias gradients; repeated indices,
    border pixels and masked slots included.  fp32 mode 2e-5 of the largest element; bf16 5e-2 (operands rounded to bf16)."""
    from centernet_amd import ops
    from centernet_amd.utils.losses import RegL1Loss
    B, Ci, Ch = 3, 64, 256
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Ci, hw, hw, generator=g)
    w1 = torch.randn(Ch, Ci, 3, 3, generator=g) * 0.05
    b1 = torch.randn(Ch, generator=g) * 0.1
    w2 = torch.randn(C, Ch, 1, 1, generator=g) * 0.1
    b2 = torch.randn(C, generator=g) * 0.1
    ind = torch.randint(0, hw * hw, (B, M), generator=g)
    ind[:, 0] = 0                       # corner pixel
    ind[:, 1] = hw * hw - 1             # opposite corner
    ind[:, 3] = ind[:, 2]               # repeated index
    mask = (torch.rand(B, M, generator=g) > 0.3)
    mask[:, :4] = True
    target = torch.randn(B, M, C, generator=g)
    if dt == torch.bfloat16:
        x, w1, w2 = x.bfloat16().float(), w1.bfloat16().float(), w2.bfloat16().float()
    # reference: a dense sibling consumer (sum of squares of a 1x1 projection) shares x with the head
    xr, w1r, b1r, w2r, b2r = (t.clone().requires_grad_() for t in (x, w1, b1, w2, b2))
    lr = _head_reference(xr, w1r, b1r, w2r, b2r, ind, mask, target) + 0.01 * (xr * xr).sum()
    lr.backward()
    # HIP: NHWC engine tensors, x shared between the head and the dense sibling
    xd = x.to(DEV).requires_grad_()
    params = [t.to(DEV).requires_grad_() for t in (w1, b1, w2, b2)]
    xn = ops.share(ops.FromNCHWFn.apply(xd, dt))
    before = ops.HeadFn.sparse_runs
    out = ops.HeadFn.apply(xn, *params)
    sib = ops.ToNCHWFn.apply(xn, Ci)
    loss = RegL1Loss()(out, mask.to(DEV), ind.to(DEV), target.to(DEV)) + 0.01 * (sib * sib).sum()
    loss.backward()
    assert ops.HeadFn.sparse_runs == before + 1, "the row path did not run"
    tol = 2e-5 if dt == torch.float32 else 5e-2
    assert float(loss) == pytest.approx(float(lr), rel=1e-5 if dt == torch.float32 else 2e-2)
    for name, got, ref in [("x", xd.grad, xr.grad), ("w1", params[0].grad, w1r.grad), ("b1", params[1].grad, b1r.grad),
                           ("w2", params[2].grad, w2r.grad), ("b2", params[3].grad, b2r.grad)]:
        err = float((got.cpu() - ref).abs().max()) / float(ref.abs().max())
        assert err < tol, (name, err)


@pytest.mark.parametrize("dense", [False, True])
def test_two_channel_head_one_launch_forward_trains(dense, monkeypatch):
    """A 2-channel bf16 head whose FORWARD is the one-launch cn_head2_fwd (no hidden activation stored): under a gather-type loss the
    backward recomputes the R hidden rows from the gathered input patches (cn_head_sparse_gather_rows modes 1 / 2 + a 1x1 GEMM); under
    a dense gradient it recomputes the hidden activation — both against torch's dense autograd of the same head."""
    from centernet_amd import ops
    from centernet_amd.utils.losses import RegL1Loss
    B, Ci, Ch, C, M, hw = 4, 64, 256, 2, 24, 32
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(B, Ci, hw, hw, generator=g)).bfloat16().float()
    w1 = (torch.randn(Ch, Ci, 3, 3, generator=g) * 0.05).bfloat16().float()
    b1 = torch.randn(Ch, generator=g) * 0.1
    w2 = (torch.randn(C, Ch, 1, 1, generator=g) * 0.1).bfloat16().float()
    b2 = torch.randn(C, generator=g) * 0.1
    ind = torch.randint(0, hw * hw, (B, M), generator=g)
    ind[:, 0], ind[:, 1], ind[:, 3] = 0, hw * hw - 1, ind[:, 2]
    mask = torch.rand(B, M, generator=g) > 0.3
    target = torch.randn(B, M, C, generator=g)
    xr, w1r, b1r, w2r, b2r = (t.clone().requires_grad_() for t in (x, w1, b1, w2, b2))
    outr = F.conv2d(F.relu(F.conv2d(xr, w1r, b1r, 1, 1)), w2r, b2r)
    lr = _head_reference(xr, w1r, b1r, w2r, b2r, ind, mask, target) + ((0.05 * (outr * outr).sum()) if dense else 0.0)
    lr.backward()
    monkeypatch.setenv("CN_CONV_WS_FORCE", "8")          # the weight-stationary kernel takes this small problem
    xd = x.to(DEV).requires_grad_()
    params = [t.to(DEV).requires_grad_() for t in (w1, b1, w2, b2)]
    xn = ops.FromNCHWFn.apply(xd, torch.bfloat16)
    before = ops.HeadFn.sparse_runs
    out = ops.HeadFn.apply(xn, *params)
    assert out.grad_fn is not None and len(out.grad_fn.saved_tensors[1].shape) == 1, "the forward stored a hidden activation"
    close(out, outr, torch.bfloat16, "one-launch head forward")
    loss = RegL1Loss()(out, mask.to(DEV), ind.to(DEV), target.to(DEV)) + ((0.05 * (out * out).sum()) if dense else 0.0)
    loss.backward()
    assert ops.HeadFn.sparse_runs == before + (0 if dense else 1)
    assert float(loss) == pytest.approx(float(lr), rel=2e-2)
    for name, got, ref in [("x", xd.grad, xr.grad), ("w1", params[0].grad, w1r.grad), ("b1", params[1].grad, b1r.grad),
                           ("w2", params[2].grad, w2r.grad), ("b2", params[3].grad, b2r.grad)]:
        err = float((got.cpu() - ref).abs().max()) / float(ref.abs().max())
        assert err < 5e-2, (name, err)


def test_head_backward_dense_fallback_matches_rows():
    """A gradient HeadFn cannot tie to a gather (here: the same map plus a dense term) takes the dense path; both paths agree."""
    from centernet_amd import ops
    from centernet_amd.utils.losses import RegL1Loss
    B, Ci, Ch, C, M, hw = 2, 64, 256, 2, 8, 16
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, Ci, hw, hw, generator=g).to(DEV)
    ws = [(torch.randn(Ch, Ci, 3, 3, generator=g) * 0.05), torch.randn(Ch, generator=g) * 0.1,
          torch.randn(C, Ch, 1, 1, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1]
    ind = torch.randint(0, hw * hw, (B, M), generator=g).to(DEV)
    mask = torch.ones(B, M, dtype=torch.bool, device=DEV)
    target = torch.randn(B, M, C, generator=g).to(DEV)
    grads = {}
    for mode in ("rows", "dense"):
        ps = [w.clone().to(DEV).requires_grad_() for w in ws]
        xd = x.clone().requires_grad_()
        out = ops.HeadFn.apply(ops.FromNCHWFn.apply(xd, torch.float32), *ps)
        before = ops.HeadFn.sparse_runs
        loss = RegL1Loss()(out, mask, ind, target) + (0.0 * out.sum() if mode == "dense" else 0.0)   # + 0 * sum: autograd adds a second gradient
        loss.backward()
        assert ops.HeadFn.sparse_runs == before + (mode == "rows")
        grads[mode] = [xd.grad] + [p.grad for p in ps]
    for a, b in zip(grads["rows"], grads["dense"]):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())


@pytest.mark.parametrize("C", [2, 34, 80])
def test_conv1x1_into_public_nchw_layout(C):
    """cn_conv1x1_nchw_fwd (a head's last conv writing the fp32 NCHW map itself) against torch's conv on the same bf16-rounded
    operands: fp32 accumulation, no bf16 rounding of the result -> 1e-5 of the largest element; and the entry point declines
    what the streaming kernel does not take (fp32 activations) so that the caller runs conv + layout change."""
    from centernet_amd import _hip, ops
    N, H, W, Ch = 4, 128, 128, 256
    g = torch.Generator().manual_seed(9)
    h = torch.randn(N, H, W, Ch, generator=g).bfloat16()
    w = (torch.randn(C, Ch, 1, 1, generator=g) * 0.1)
    b = torch.randn(C, generator=g)
    ref = torch.nn.functional.conv2d(h.float().permute(0, 3, 1, 2), w.bfloat16().float(), b)
    hd, wd, bd = h.to(DEV), w.to(DEV), b.to(DEV)
    wp = ops.pack_weight(wd, 1, torch.bfloat16)
    out = torch.empty((N, C, H, W), dtype=torch.float32, device=DEV)
    assert _hip.try_call("cn_conv1x1_nchw_fwd", hd, wp, bd, out, N, H, W, Ch, Ch, C, _hip.CN_BF16)
    err = float((out.cpu() - ref).abs().max()) / float(ref.abs().max())
    assert err < 1e-5, err
    assert torch.equal(ops.conv1x1_to_nchw(hd, wp, bd, C), out)
    h32 = hd.float()
    assert not _hip.try_call("cn_conv1x1_nchw_fwd", h32, ops.pack_weight(wd, 1, torch.float32), bd, out, N, H, W, Ch, Ch, C, _hip.CN_F32)
    out32 = ops.conv1x1_to_nchw(h32, ops.pack_weight(wd, 1, torch.float32), bd, C)      # general path
    assert float((out32.cpu() - torch.nn.functional.conv2d(h.float().permute(0, 3, 1, 2), w, b)).abs().max()) < 1e-3 * float(ref.abs().max())
