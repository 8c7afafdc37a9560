"""A/B of conv3x3_kp variants on ONE box: python tools/attic/kp_bench.py lib1.so lib2.so ...   ("old" = the default library with
CN_DISABLE_CONV_KP=1, "cur" = the default library).  Each variant runs in its own process, three interleaved rounds."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, ROOT)
    from centernet_amd import ops
    out = []
    for Ci, HW in ((128, 64), (256, 32), (512, 16)):
        x = torch.randn(64, HW, HW, Ci, device="cuda").bfloat16()
        w = torch.randn(Ci, Ci, 3, 3, device="cuda") * (2.0 / (9 * Ci)) ** 0.5
        wp = ops.pack_weight(w, 1, torch.bfloat16)
        fn = lambda: ops._igemm(x, wp, None, None, Ci, 3, 3, 1, 1, False, False, HW, HW)
        for _ in range(5): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        fl = 2.0 * 64 * HW * HW * 9 * Ci * Ci
        out.append(f"{Ci}@{HW}: {ts[len(ts) // 2]:6.1f} us {fl / ts[len(ts) // 2] / 2.5e9 * 100:4.1f}%")
    print("   ".join(out))
else:
    for rnd in range(3):
        for v in sys.argv[1:]:
            env = dict(os.environ, CN_ENABLE_CONV_KP="1")
            if v == "old": env["CN_DISABLE_CONV_KP"] = "1"
            elif v == "stagger": env["CN_CONV_KP_MODE"] = "stagger"
            elif v != "cur": env["CN_LIB_PATH"] = os.path.abspath(v)
            r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
            print(f"round {rnd} {os.path.basename(v):22s} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
