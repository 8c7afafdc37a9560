"""Development aid for dcn_dom_bm_kernel: builds the library with -DDOMB_PROBE into tools/_ab/, runs one 64->64 @128^2 launch and
prints the median cycles of its phases.   python tools/dom_probe.py build | run [sigma]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "_ab", "lib_dom_probe.so")
CSRC = os.path.join(ROOT, "centernet-pytorch-lightning_amd", "csrc")
if sys.argv[1] == "build":
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    objs = [os.path.join(CSRC, "_build", f) for f in sorted(os.listdir(os.path.join(CSRC, "_build"))) if f.endswith(".o") and f != "dcn_dom_bm.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DDOMB_PROBE"] + ([os.environ["DOMB_EXTRA"]] if os.environ.get("DOMB_EXTRA") else []) + ["-c",
                           os.path.join(CSRC, "dcn_dom_bm.hip"), "-o", "/tmp/dom_probe.o"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, "/tmp/dom_probe.o"] + objs)
    print("built", SO)
else:
    os.environ["CN_LIB_PATH"] = SO
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from centernet_amd import _hip, ops
    sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    N, H, W, Ci, Co = 64, 128, 128, 64, 64
    dt = torch.bfloat16
    x = torch.randn(N, H, W, Ci, device="cuda").to(dt)
    dy = torch.randn(N, H, W, Co, device="cuda").to(dt)
    om = torch.zeros(N, H, W, 32, device="cuda")
    if sigma:
        om[..., :18] = torch.randn(N, H, W, 18, device="cuda") * sigma
        om[..., 18:27] = torch.randn(N, H, W, 9, device="cuda")
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.04
    wp = ops.pack_weight(w, 2, dt)
    dom = torch.empty(N, H, W, 32, device="cuda", dtype=dt)
    far = torch.zeros(N, H, W, Ci, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    code = _hip.dtype_code(dt)
    run = lambda: _hip.call("cn_dcn_bwd_dom", dy, wp, x, om, dom, 0, far, flag, N, H, W, Ci, Co, Co, Ci, 32, code)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    lib = ctypes.CDLL(SO)
    buf = np.zeros(1024 * 48, dtype=np.uint64)
    assert lib.domb_probe_dump(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    ts = buf.reshape(1024, 48).astype(np.int64)
    ts = ts[ts[:, 39] != 0]
    print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us, offsets sigma {sigma}, {len(ts)} workgroups stamped")
    med = lambda v: f"median {np.median(v):8.0f}  p10 {np.percentile(v, 10):8.0f}  p90 {np.percentile(v, 90):8.0f}"
    print("  prologue (start -> barrier)      ", med(ts[:, 1] - ts[:, 0]))
    print("    start -> geometry table written", med(ts[:, 40] - ts[:, 0]))
    print("    halo image registers -> LDS    ", med(ts[:, 41] - ts[:, 40]))
    print("    W store + barrier              ", med(ts[:, 1] - ts[:, 41]))
    a = np.stack([ts[:, 3 + 4 * t] - ts[:, 2 + 4 * t] for t in range(9)], 1)
    b = np.stack([ts[:, 4 + 4 * t] - ts[:, 3 + 4 * t] for t in range(9)], 1)
    c = np.stack([ts[:, 5 + 4 * t] - ts[:, 4 + 4 * t] for t in range(9)], 1)
    d = np.stack([(ts[:, 6 + 4 * t] if t < 8 else ts[:, 38]) - ts[:, 5 + 4 * t] for t in range(9)], 1)
    print("  per tap: geometry + dcol + W load", med(a), " by tap:", np.median(a, 0).astype(int))
    print("  per tap: row-pair loop           ", med(b), " by tap:", np.median(b, 0).astype(int))
    print("  per tap: rare-path test          ", med(c))
    print("  per tap: combine + park          ", med(d))
    print("  nine taps                        ", med(ts[:, 38] - ts[:, 1]))
    print("  flush                            ", med(ts[:, 39] - ts[:, 38]))
    print("  whole workgroup                  ", med(ts[:, 39] - ts[:, 0]))
    span = ts[:, 39].max() - ts[:, 0].min()
    print(f"  span of the stamped workgroups: {span} ticks")
