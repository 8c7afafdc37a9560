"""Development aid for dcn_bwd_dx_kernel: builds the library with -DDX_PROBE into tools/_ab/, runs one 64->64 @128^2 launch and
prints the median cycles of the three phases of a tap (hit lists | G tile | MFMA).   python tools/attic/dx_probe.py build | run [sigma]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "_ab", "lib_dx_probe.so")
CSRC = os.path.join(ROOT, "centernet-pytorch-lightning_amd", "csrc")
if sys.argv[1] == "build":
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    objs = [os.path.join(CSRC, "_build", f) for f in sorted(os.listdir(os.path.join(CSRC, "_build"))) if f.endswith(".o") and f != "dcn_fused.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DDX_PROBE", "-c",
                           os.path.join(CSRC, "dcn_fused.hip"), "-o", "/tmp/dx_probe.o"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, "/tmp/dx_probe.o"] + objs)
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
    wp0 = ops.pack_weight(w, 0, dt)
    far = ops._far_buffer((N, H, W, Ci), "cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    dx = torch.empty(N, H, W, Ci, device="cuda", dtype=dt)
    code = _hip.dtype_code(dt)
    for _ in range(3):
        _hip.call("cn_dcn_bwd_dx", dy, wp0, om, far, flag, dx, N, H, W, Ci, Co, 32, code)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _hip.call("cn_dcn_bwd_dx", dy, wp0, om, far, flag, dx, N, H, W, Ci, Co, 32, code); e1.record()
    torch.cuda.synchronize()
    lib = ctypes.CDLL(SO)
    buf = np.zeros(1024 * 9 * 4, dtype=np.uint64)
    assert lib.dx_probe_dump(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    ts = buf.reshape(1024, 9, 4).astype(np.int64)[:128]
    print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us, offsets sigma {sigma}")
    for nm, a, b in (("hit lists", 0, 1), ("G tile + weight store", 1, 2), ("MFMA", 2, 3)):
        v = ts[:, :, b] - ts[:, :, a]
        print(f"  {nm:24s} median {np.median(v):8.0f}  p10 {np.percentile(v, 10):8.0f}  p90 {np.percentile(v, 90):8.0f} cycles")
    print("  tap period", np.median(ts[:, 1:, 0] - ts[:, :-1, 0]), "cycles; workgroup", np.median(ts[:, 8, 3] - ts[:, 0, 0]))
