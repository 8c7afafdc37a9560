"""Development check of the blend-matrix DCNv2 kernels (csrc/dcn_bm.hip): cn_dcn_fwd on random inputs at several offset scales
(zero, sub-pixel, > 3 px = the far fallback) against oracle/dcn_ref.py on the CPU, plus an isolated timing at the bench shape.
    python tools/dcn_bm_check.py            # uses whatever cn_dcn_fwd dispatches to (CN_DISABLE_DCN_BM=1: the gather kernel)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centernet_amd import _hip, ops  # noqa: E402
from oracle.dcn_ref import dcn_v2_conv  # noqa: E402

DEV = "cuda"


def check(N, H, W, Ci, Co, sigma, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Ci, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * (2.0 / (9 * Ci)) ** 0.5).bfloat16().float()
    bias = torch.randn(Co, generator=g) * 0.1
    off = torch.randn(N, 18, H, W, generator=g) * sigma
    ml = torch.randn(N, 9, H, W, generator=g)
    ref = dcn_v2_conv(x, off, torch.sigmoid(ml), w, bias)
    om = torch.zeros(N, H, W, 32)
    om[..., :18] = off.permute(0, 2, 3, 1)
    om[..., 18:27] = ml.permute(0, 2, 3, 1)
    dt = torch.bfloat16
    xg = x.permute(0, 2, 3, 1).contiguous().to(DEV).to(dt)
    wp = ops.pack_weight(w.to(DEV), 1, dt)
    y = torch.empty(N, H, W, Co, device=DEV, dtype=dt)
    _hip.call("cn_dcn_fwd", xg, om.to(DEV), wp, bias.to(DEV), y, N, H, W, Ci, Ci, Co, Co, 32, 0, _hip.dtype_code(dt))
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs().max().item()
    rms = ((got - ref) ** 2).mean().sqrt().item()
    sc = ref.abs().max().item()
    print(f"N{N} {H}x{W} {Ci}->{Co} sigma {sigma:4.1f}: max err {err:.4f} ({err / sc:.2e} of max {sc:.2f}), rms {rms:.5f} ({rms / ref.std().item():.2e} of std)", flush=True)
    return err / sc


def check_dx(N, H, W, Ci, Co, sigma, seed=0):
    """sampling-path data gradient (offsets held constant) through cn_dcn_bwd_dom (far samples) + cn_dcn_bwd_dx vs autograd of the oracle"""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Ci, H, W, generator=g).bfloat16().float().requires_grad_(True)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * (2.0 / (9 * Ci)) ** 0.5).bfloat16().float()
    off = torch.randn(N, 18, H, W, generator=g) * sigma
    ml = torch.randn(N, 9, H, W, generator=g)
    gy = torch.randn(N, Co, H, W, generator=g).bfloat16().float()
    dcn_v2_conv(x, off, torch.sigmoid(ml), w, None).backward(gy)
    ref = x.grad
    om = torch.zeros(N, H, W, 32)
    om[..., :18] = off.permute(0, 2, 3, 1)
    om[..., 18:27] = ml.permute(0, 2, 3, 1)
    dt = torch.bfloat16
    code = _hip.dtype_code(dt)
    xg = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV).to(dt)
    dy = gy.permute(0, 2, 3, 1).contiguous().to(DEV).to(dt)
    omg = om.to(DEV)
    wd = w.to(DEV)
    far = torch.zeros(N, H, W, Ci, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    slabs = _hip.query("cn_dcn_bwd_dom_slabs", Ci, Co, code)
    dom = torch.zeros(max(slabs, 1), N, H, W, 32, device=DEV)
    _hip.call("cn_dcn_bwd_dom", dy, ops.pack_weight(wd, 2, dt), xg, omg, dom, slabs, far, flag, N, H, W, Ci, Co, Co, Ci, 32, code)
    dx = torch.empty(N, H, W, Ci, device=DEV, dtype=dt)
    _hip.call("cn_dcn_bwd_dx", dy, ops.pack_weight(wd, 0, dt), omg, far, flag, dx, N, H, W, Ci, Co, 32, code)
    torch.cuda.synchronize()
    got = dx.float().cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs().max().item()
    rms = ((got - ref) ** 2).mean().sqrt().item()
    sc = ref.abs().max().item()
    assert float(far.abs().max()) == 0.0, "dx_far must be left clean"
    print(f"dx  N{N} {H}x{W} {Ci}<-{Co} sigma {sigma:4.1f}: max err {err:.4f} ({err / sc:.2e} of max {sc:.2f}), rms {rms:.5f} ({rms / ref.std().item():.2e} of std), far flag {int(flag.item())}", flush=True)
    return err / sc


def check_dw(N, H, W, sigma, seed=0, Ci=64, Co=64):
    """weight gradient of the deformable conv through cn_dcn_wgrad vs autograd of the oracle"""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Ci, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * (2.0 / (9 * Ci)) ** 0.5).bfloat16().float().requires_grad_(True)
    off = torch.randn(N, 18, H, W, generator=g) * sigma
    ml = torch.randn(N, 9, H, W, generator=g)
    gy = torch.randn(N, Co, H, W, generator=g).bfloat16().float()
    dcn_v2_conv(x, off, torch.sigmoid(ml), w, None).backward(gy)
    ref = w.grad
    om = torch.zeros(N, H, W, 32)
    om[..., :18] = off.permute(0, 2, 3, 1)
    om[..., 18:27] = ml.permute(0, 2, 3, 1)
    dt = torch.bfloat16
    code = _hip.dtype_code(dt)
    xg = x.permute(0, 2, 3, 1).contiguous().to(DEV).to(dt)
    dy = gy.permute(0, 2, 3, 1).contiguous().to(DEV).to(dt)
    dwp = torch.zeros(Co, 9 * Ci, device=DEV)
    _hip.call("cn_dcn_wgrad", xg, om.to(DEV), dy, dwp, N, H, W, Ci, Ci, Co, Co, 32, code)
    torch.cuda.synchronize()
    got = dwp.cpu().view(Co, 9, Ci).permute(0, 2, 1).reshape(Co, Ci, 3, 3)
    err = (got - ref).abs().max().item()
    rms = ((got - ref) ** 2).mean().sqrt().item()
    sc = ref.abs().max().item()
    print(f"dW  N{N} {H}x{W} {Ci}->{Co} sigma {sigma:4.1f}: max err {err:.4f} ({err / sc:.2e} of max {sc:.2f}), rms {rms:.5f} ({rms / ref.std().item():.2e} of std)", flush=True)
    return err / sc


def check_dom(N, H, W, Ci, sigma, seed=0):
    """offset / mask gradient (cn_dcn_bwd_dom, dY with 64 channels) vs autograd of the oracle w.r.t. the offsets and mask logits"""
    Co = 64
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Ci, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * (2.0 / (9 * Ci)) ** 0.5).bfloat16().float()
    off = (torch.randn(N, 18, H, W, generator=g) * sigma).requires_grad_(True)
    ml = torch.randn(N, 9, H, W, generator=g).requires_grad_(True)
    gy = torch.randn(N, Co, H, W, generator=g).bfloat16().float()
    dcn_v2_conv(x, off, torch.sigmoid(ml), w, None).backward(gy)
    ref = torch.cat([off.grad, ml.grad], 1).permute(0, 2, 3, 1)           # [N,H,W,27]
    om = torch.zeros(N, H, W, 32)
    om[..., :18] = off.detach().permute(0, 2, 3, 1)
    om[..., 18:27] = ml.detach().permute(0, 2, 3, 1)
    dt = torch.bfloat16
    code = _hip.dtype_code(dt)
    xg = x.permute(0, 2, 3, 1).contiguous().to(DEV).to(dt)
    dy = gy.permute(0, 2, 3, 1).contiguous().to(DEV).to(dt)
    far = torch.zeros(N, H, W, Ci, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    slabs = _hip.query("cn_dcn_bwd_dom_slabs", Ci, Co, code)
    dom = torch.full((max(slabs, 1), N, H, W, 32), float("nan"), device=DEV) if slabs == Ci // 64 else torch.zeros(1, N, H, W, 32, device=DEV)
    _hip.call("cn_dcn_bwd_dom", dy, ops.pack_weight(w.to(DEV), 2, dt), xg, om.to(DEV), dom, slabs if slabs == Ci // 64 else 1, far, flag,
              N, H, W, Ci, Co, Co, Ci, 32, code)
    torch.cuda.synchronize()
    got = dom.sum(0).cpu()[..., :27]
    assert float(dom.sum(0)[..., 27:].abs().max()) == 0.0, "channel padding must be written as zeros"
    err = (got - ref).abs().max().item()
    rms = ((got - ref) ** 2).mean().sqrt().item()
    sc = ref.abs().max().item()
    print(f"dom N{N} {H}x{W} Ci {Ci} sigma {sigma:4.1f}: max err {err:.4f} ({err / sc:.2e} of max {sc:.2f}), rms {rms:.5f} ({rms / ref.std().item():.2e} of std), far flag {int(flag.item())}", flush=True)
    return err / sc


if __name__ == "__main__":
    print("CN_DISABLE_DCN_BM =", os.environ.get("CN_DISABLE_DCN_BM"))
    worst = 0.0
    for (N, H, W, Co) in [(2, 16, 32, 64), (1, 13, 21, 64), (1, 8, 16, 32)]:
        for sigma in (0.0, 0.5, 1.5, 4.0):
            worst = max(worst, check(N, H, W, 64, Co, sigma))
    for Ci in (128, 256):                              # channel blocks of x one after the other (round 5)
        for sigma in (0.0, 1.5, 4.0):
            worst = max(worst, check(2, 16, 32, Ci, 64, sigma))
    print("worst rel err", worst)
    worst = 0.0
    for (N, H, W, Ci) in [(2, 16, 32, 64), (1, 13, 21, 64), (1, 8, 16, 32)]:
        for sigma in (0.0, 0.5, 1.5, 4.0):
            worst = max(worst, check_dx(N, H, W, Ci, 64, sigma))
    for Ci in (128, 256):
        for sigma in (0.0, 1.5, 4.0):
            worst = max(worst, check_dx(2, 16, 32, Ci, 64, sigma))
    print("worst rel err dx", worst)
    worst = 0.0
    for (N, H, W) in [(2, 16, 32), (1, 13, 21), (3, 24, 48)]:
        for sigma in (0.0, 0.5, 1.5, 4.0):
            worst = max(worst, check_dw(N, H, W, sigma))
    for Ci, Co in [(128, 64), (128, 128), (256, 64)]:
        worst = max(worst, check_dw(2, 16, 32, 1.5, Ci=Ci, Co=Co))
    print("worst rel err dW", worst)
    worst = 0.0
    for (N, H, W, Ci) in [(2, 16, 32, 64), (1, 13, 21, 64), (2, 16, 16, 128)]:
        for sigma in (0.0, 0.5, 1.5, 4.0):
            worst = max(worst, check_dom(N, H, W, Ci, sigma))
    print("worst rel err dom", worst)
    if len(sys.argv) > 1 and sys.argv[1] == "time":
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import opbench
        os.environ["DCN_SHAPES"] = "1"
        opbench.bench_dcn()
