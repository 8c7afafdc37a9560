"""Development check + A/B timing of dcn_fwd_b2_kernel (csrc/dcn_b2.hip) against the oracle and against dcn_fwd_bm_kernel.
    python tools/dev/b2_fwd.py            (spawns itself with / without CN_DISABLE_DCN_FWD_B2 for the timing)"""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import dcn_bm_check as c
    import opbench
    from centernet_amd import _hip, ops
    mode = sys.argv[1]
    if mode == "check":
        worst = 0.0
        for (N, H, W, Co) in [(2, 16, 32, 64), (1, 13, 21, 64), (1, 37, 50, 64), (2, 64, 64, 64)]:
            for sigma in (0.0, 0.5, 1.5, 4.0):
                worst = max(worst, c.check(N, H, W, 64, Co, sigma))
        for Ci in (128, 256):
            for sigma in (0.0, 1.5, 4.0):
                worst = max(worst, c.check(2, 16, 32, Ci, 64, sigma))
        print("worst rel err", worst)
    else:
        dt = torch.bfloat16
        code = _hip.dtype_code(dt)
        for HW, Ci, Co in [(128, 64, 64), (64, 128, 64), (32, 256, 64)]:
            N, H, W = 64, HW, HW
            for tag, sigma in (("zero", 0.0), ("N(0,0.5)", 0.5)):
                x = torch.randn(N, H, W, Ci, device="cuda").to(dt)
                om = torch.zeros(N, H, W, 32, device="cuda")
                if sigma:
                    om[..., :18] = torch.randn(N, H, W, 18, device="cuda") * sigma
                    om[..., 18:27] = torch.randn(N, H, W, 9, device="cuda")
                w = (torch.randn(Co, Ci, 3, 3) * (2.0 / (9 * Ci)) ** 0.5).cuda()
                bias = torch.zeros(Co, device="cuda")
                wp1 = ops.pack_weight(w, 1, dt)
                y = torch.empty(N, H, W, Co, device="cuda", dtype=dt)
                us, mn = opbench.timeit(lambda: _hip.call("cn_dcn_fwd", x, om, wp1, bias, y, N, H, W, Ci, Ci, Co, Co, 32, 0, code), n=20)
                print(f"{mode:4s} dcn fwd {Ci:3d}->{Co:3d} @{HW:3d}^2 [{tag:9s}] {us:8.1f} us (min {mn:8.1f})", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        main()
    else:
        env = dict(os.environ, CN_DCN_B2_MIN_HW="0")
        subprocess.run([sys.executable, __file__, "check"], env=env)
        subprocess.run([sys.executable, __file__, "b2"], env=dict(os.environ, CN_DCN_B2_MIN_HW="0"))
        subprocess.run([sys.executable, __file__, "bm"], env=dict(os.environ, CN_DISABLE_DCN_FWD_B2="1"))
