"""Development aid for dcn_dx_bm_kernel: builds the library with -DDXB_PROBE into tools/_ab/, runs one 64->64 @128^2 launch and
prints the median cycles of its phases.   python tools/dxbm_probe.py build | run [sigma]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "_ab", "lib_dxbm_probe.so")
CSRC = os.path.join(ROOT, "centernet-pytorch-lightning_amd", "csrc")
if sys.argv[1] == "build":
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    objs = [os.path.join(CSRC, "_build", f) for f in sorted(os.listdir(os.path.join(CSRC, "_build"))) if f.endswith(".o") and f != "dcn_bm.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DDXB_PROBE", "-c",
                           os.path.join(CSRC, "dcn_bm.hip"), "-o", "/tmp/dxbm_probe.o"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, "/tmp/dxbm_probe.o"] + objs)
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
    dy = torch.randn(N, H, W, Co, device="cuda").to(dt)
    om = torch.zeros(N, H, W, 32, device="cuda")
    if sigma:
        om[..., :18] = torch.randn(N, H, W, 18, device="cuda") * sigma
        om[..., 18:27] = torch.randn(N, H, W, 9, device="cuda")
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.04
    wp = ops.pack_weight(w, 0, dt)
    dx = torch.empty(N, H, W, Ci, device="cuda", dtype=dt)
    far = torch.zeros(N, H, W, Ci, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    code = _hip.dtype_code(dt)
    run = lambda: _hip.call("cn_dcn_bwd_dx", dy, wp, om, far, flag, dx, N, H, W, Ci, Co, 32, code)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    lib = ctypes.CDLL(SO)
    buf = np.zeros(1024 * 40, dtype=np.uint64)
    assert lib.dxb_probe_dump(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    ts = buf.reshape(1024, 40).astype(np.int64)
    ts = ts[ts[:, 31] != 0]
    print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us, offsets sigma {sigma}, {len(ts)} workgroups stamped")
    med = lambda v: f"median {np.median(v):8.0f}  p10 {np.percentile(v, 10):8.0f}  p90 {np.percentile(v, 90):8.0f}"
    print("  prologue (halo, fragments, table)", med(ts[:, 1] - ts[:, 0]))
    print("    loads -> halo image in LDS      ", med(ts[:, 32] - ts[:, 0]))
    print("    window fragments                ", med(ts[:, 33] - ts[:, 32]))
    print("    geometry table                  ", med(ts[:, 1] - ts[:, 33]))
    a = np.stack([ts[:, 3 + 3 * t] - ts[:, 2 + 3 * t] for t in range(9)], 1)
    b = np.stack([ts[:, 4 + 3 * t] - ts[:, 3 + 3 * t] for t in range(9)], 1)
    c = np.stack([(ts[:, 5 + 3 * t] if t < 8 else ts[:, 29]) - ts[:, 4 + 3 * t] for t in range(9)], 1)
    print("  per tap: three T-tile passes      ", med(a), " by tap:", np.median(a, 0).astype(int))
    print("  per tap: contraction              ", med(b))
    print("  per tap: W stage + barrier        ", med(c))
    print("  nine taps                         ", med(ts[:, 29] - ts[:, 1]))
    print("  epilogue                          ", med(ts[:, 31] - ts[:, 29]))
    print("  whole workgroup                   ", med(ts[:, 31] - ts[:, 0]))
