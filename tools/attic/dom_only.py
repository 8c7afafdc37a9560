"""Development check of dcn_dom_bm_kernel: cn_dcn_bwd_dom against autograd of the oracle, with the location of the worst element."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from centernet_amd import _hip, ops  # noqa: E402
from oracle.dcn_ref import dcn_v2_conv  # noqa: E402

DEV = "cuda"


def run(N, H, W, Ci, Co, sigma, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Ci, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * (2.0 / (9 * Ci)) ** 0.5).bfloat16().float()
    off = (torch.randn(N, 18, H, W, generator=g) * sigma).requires_grad_(True)
    ml = torch.randn(N, 9, H, W, generator=g).requires_grad_(True)
    gy = torch.randn(N, Co, H, W, generator=g).bfloat16().float()
    dcn_v2_conv(x, off, torch.sigmoid(ml), w, None).backward(gy)
    ref = torch.cat([off.grad, ml.grad], 1).permute(0, 2, 3, 1)
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
    dom = torch.full((max(slabs, 1), N, H, W, 32), float("nan"), device=DEV)
    _hip.call("cn_dcn_bwd_dom", dy, ops.pack_weight(w.to(DEV), 2, dt), xg, om.to(DEV), dom, slabs, far, flag, N, H, W, Ci, Co, Co, Ci, 32, code)
    torch.cuda.synchronize()
    got = dom.sum(0).cpu()[..., :27]
    d = (got - ref).abs()
    sc = ref.abs().max().item()
    i = int(d.argmax())
    n, h, ww, c = [int(v) for v in torch.unravel_index(torch.tensor(i), d.shape)]
    k = c // 2 if c < 18 else c - 18
    print(f"dom N{N} {H}x{W} {Ci}->{Co} sigma {sigma}: max err {d.max().item():.4f} ({d.max().item() / sc:.2e}) at n{n} h{h} w{ww} c{c} (tap {k}): got {got[n, h, ww, c]:.4f} ref {ref[n, h, ww, c]:.4f}"
          f" | offsets dy {om[n, h, ww, 2 * k]:.3f} dx {om[n, h, ww, 2 * k + 1]:.3f} -> sample ({h - 1 + k // 3 + om[n, h, ww, 2 * k]:.3f}, {ww - 1 + k % 3 + om[n, h, ww, 2 * k + 1]:.3f})"
          f" far-sum {float(far.abs().sum()):.3f}", flush=True)
    return d.max().item() / sc


if __name__ == "__main__":
    worst = 0.0
    for (N, H, W, Ci, Co) in [(2, 16, 32, 64, 64), (1, 13, 21, 64, 64), (2, 16, 16, 128, 64), (3, 24, 48, 64, 64), (2, 16, 32, 64, 128), (1, 24, 40, 128, 128)]:
        for sigma in (0.0, 0.5, 1.5, 4.0):
            worst = max(worst, run(N, H, W, Ci, Co, sigma))
    for seed in range(1, 6):
        worst = max(worst, run(3, 24, 48, 64, 64, 4.0, seed))
    print("worst rel err dom", worst)
