"""Phase stamps of dcn_fwd_b2_kernel: python tools/dev/b2_probe.py build [wave] | run [sigma]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SO = os.path.join(ROOT, "tools", "_ab", "lib_b2_probe.so")
if sys.argv[1] == "build":
    wave = sys.argv[2] if len(sys.argv) > 2 else "0"
    subprocess.check_call(["bash", os.path.join(ROOT, "tools/_ab/build_variant.sh"), "b2_probe", "dcn_b2.hip", f"-DB2_PROBE -DB2_PROBE_WAVE={wave}"])
else:
    os.environ["CN_LIB_PATH"] = SO
    os.environ["CN_DCN_B2_MIN_HW"] = "0"
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    from centernet_amd import _hip, ops
    sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    N, H, W, Ci, Co = 64, 128, 128, 64, 64
    dt = torch.bfloat16
    x = torch.randn(N, H, W, Ci, device="cuda").to(dt)
    om = torch.zeros(N, H, W, 32, device="cuda")
    if sigma:
        om[..., :18] = torch.randn(N, H, W, 18, device="cuda") * sigma
        om[..., 18:27] = torch.randn(N, H, W, 9, device="cuda")
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.04
    wp = ops.pack_weight(w, 1, dt); bias = torch.zeros(Co, device="cuda")
    y = torch.empty(N, H, W, Co, device="cuda", dtype=dt); code = _hip.dtype_code(dt)
    run = lambda: _hip.call("cn_dcn_fwd", x, om, wp, bias, y, N, H, W, Ci, Ci, Co, Co, 32, 0, code)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    lib = ctypes.CDLL(SO)
    buf = np.zeros(2048 * 48, dtype=np.uint64)
    assert lib.b2_probe_dump(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    ts = buf.reshape(2048, 48).astype(np.int64)
    ts = ts[ts[:, 39] != 0]
    print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us, sigma {sigma}, {len(ts)} workgroups stamped")
    med = lambda v: f"median {np.median(v):8.0f}  p10 {np.percentile(v, 10):8.0f}  p90 {np.percentile(v, 90):8.0f}"
    print("  prologue (start -> halo barrier)   ", med(ts[:, 1] - ts[:, 0]))
    geo = np.stack([ts[:, 3 + 4 * t] - ts[:, 2 + 4 * t] for t in range(9)], 1)
    bl = np.stack([ts[:, 4 + 4 * t] - ts[:, 3 + 4 * t] for t in range(9)], 1)
    ct = np.stack([ts[:, 5 + 4 * t] - ts[:, 4 + 4 * t] for t in range(9)], 1)
    bt = np.stack([(ts[:, 6 + 4 * t] if t < 8 else ts[:, 38]) - ts[:, 5 + 4 * t] for t in range(9)], 1)
    print("  per tap: issue + geometry + rowmask", med(geo))
    print("  per tap: blend (rows)              ", med(bl))
    print("  per tap: far + contraction         ", med(ct))
    print("  per tap: waits + W store + barrier ", med(bt))
    print("  per tap total                      ", med(geo + bl + ct + bt))
    print("  epilogue                           ", med(ts[:, 39] - ts[:, 38]))
    print("  whole workgroup                    ", med(ts[:, 39] - ts[:, 0]))
    st = ts[:, 0] - ts[:, 0].min()
    print("  start times: first 512 workgroups  ", med(st[:512]), " later:", med(st[512:1024]) if len(st) > 600 else "")
