"""Development aid for dcn_fwd_bm_kernel: builds the library with -DBM_PROBE into tools/_ab/, runs one 64->64 @128^2 launch and
prints the median cycles of its phases.   python tools/bm_probe.py build | run [sigma]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "_ab", "lib_bm_probe.so")
CSRC = os.path.join(ROOT, "centernet-pytorch-lightning_amd", "csrc")
if sys.argv[1] == "build":
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    objs = [os.path.join(CSRC, "_build", f) for f in sorted(os.listdir(os.path.join(CSRC, "_build"))) if f.endswith(".o") and f != "dcn_bm.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DBM_PROBE", "-c",
                           os.path.join(CSRC, "dcn_bm.hip"), "-o", "/tmp/bm_probe.o"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, "/tmp/bm_probe.o"] + objs)
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
    om = torch.zeros(N, H, W, 32, device="cuda")
    if sigma:
        om[..., :18] = torch.randn(N, H, W, 18, device="cuda") * sigma
        om[..., 18:27] = torch.randn(N, H, W, 9, device="cuda")
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.04
    wp = ops.pack_weight(w, 1, dt)
    bias = torch.zeros(Co, device="cuda")
    y = torch.empty(N, H, W, Co, device="cuda", dtype=dt)
    code = _hip.dtype_code(dt)
    run = lambda: _hip.call("cn_dcn_fwd", x, om, wp, bias, y, N, H, W, Ci, Ci, Co, Co, 32, 0, code)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    lib = ctypes.CDLL(SO)
    buf = np.zeros(1024 * 32, dtype=np.uint64)
    assert lib.bm_probe_dump(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    ts = buf.reshape(1024, 32).astype(np.int64)
    ts = ts[ts[:, 31] != 0]
    print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us, offsets sigma {sigma}   (clock64 ticks: 100 MHz s_memtime? check ratio below)")
    med = lambda v: f"median {np.median(v):8.0f}  p10 {np.percentile(v, 10):8.0f}  p90 {np.percentile(v, 90):8.0f}"
    print("  staging (start -> halo barrier)  ", med(ts[:, 1] - ts[:, 0]))
    print("  fragment loads                   ", med(ts[:, 2] - ts[:, 1]))
    print("  frag barrier -> tap 0            ", med(ts[:, 4] - ts[:, 2]))
    geo = np.stack([ts[:, 5 + 3 * t] - ts[:, 4 + 3 * t] for t in range(9)], 1)
    m1 = np.stack([ts[:, 6 + 3 * t] - ts[:, 5 + 3 * t] for t in range(9)], 1)
    m2 = np.stack([(ts[:, 7 + 3 * t] if t < 8 else ts[:, 3]) - ts[:, 6 + 3 * t] for t in range(9)], 1)
    print("  per tap: geometry + selectors    ", med(geo))
    print("  per tap: blend MFMAs             ", med(m1))
    print("  per tap: contraction + W stage   ", med(m2))
    print("  epilogue                         ", med(ts[:, 31] - ts[:, 3]))
    print("  whole workgroup                  ", med(ts[:, 31] - ts[:, 0]))
