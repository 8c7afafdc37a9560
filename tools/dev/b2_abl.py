"""ablation timing of dcn_fwd_b2_kernel (lib built with -DB2_ABL): CN_B2_ABL bit mask 1 = no om loads, 2 = no W staging / barrier,
4 = no contraction MFMAs, 8 = no blend"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch, opbench
    from centernet_amd import _hip, ops
    dt = torch.bfloat16; code = _hip.dtype_code(dt)
    N, H, W, Ci, Co = 64, 128, 128, 64, 64
    x = torch.randn(N, H, W, Ci, device="cuda").to(dt)
    om = torch.zeros(N, H, W, 32, device="cuda")
    om[..., :18] = torch.randn(N, H, W, 18, device="cuda") * 0.5
    om[..., 18:27] = torch.randn(N, H, W, 9, device="cuda")
    w = (torch.randn(Co, Ci, 3, 3) * (2.0 / (9 * Ci)) ** 0.5).cuda()
    bias = torch.zeros(Co, device="cuda"); wp1 = ops.pack_weight(w, 1, dt)
    y = torch.empty(N, H, W, Co, device="cuda", dtype=dt)
    us, mn = opbench.timeit(lambda: _hip.call("cn_dcn_fwd", x, om, wp1, bias, y, N, H, W, Ci, Ci, Co, Co, 32, 0, code), n=20)
    print(f"abl {sys.argv[1]:>3s}: {us:8.1f} us (min {mn:8.1f})", flush=True)
else:
    for a in sys.argv[2:] if len(sys.argv) > 2 else ["0", "1", "2", "4", "8", "12", "3", "15", "0"]:
        subprocess.run([sys.executable, __file__, a], env=dict(os.environ, CN_B2_ABL=a, CN_LIB_PATH=os.path.join(ROOT, "tools/_ab/lib_b2abl.so")))
