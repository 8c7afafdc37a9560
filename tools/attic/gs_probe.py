"""Development aid for dcn_fwd_gs_kernel: builds the library with -DGS_PROBE into tools/_ab/, runs one 64->64 @128^2 launch and prints
the median cycles of the phases of a tile (waves 0 = role 0 and 1 = role 1 of the first 64 workgroups, their first 4 tiles).
    python tools/attic/gs_probe.py build | run [sigma]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "_ab", "lib_gs_probe.so")
CSRC = os.path.join(ROOT, "centernet-pytorch-lightning_amd", "csrc")
if sys.argv[1] == "build":
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    objs = [os.path.join(CSRC, "_build", f) for f in sorted(os.listdir(os.path.join(CSRC, "_build"))) if f.endswith(".o") and f != "dcn_gs.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DGS_PROBE"] + ([os.environ["GS_EXTRA"]] if os.environ.get("GS_EXTRA") else []) + ["-c",
                           os.path.join(CSRC, "dcn_gs.hip"), "-o", "/tmp/gs_probe.o"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, "/tmp/gs_probe.o"] + objs)
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
    buf = np.zeros(64 * 3 * 4 * 16, dtype=np.uint64)
    assert lib.gs_probe_dump(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    ts = buf.reshape(64, 3, 4, 16).astype(np.int64)
    print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us, offsets sigma {sigma}")
    med = lambda v: f"median {np.median(v):8.0f}  p10 {np.percentile(v, 10):8.0f}  p90 {np.percentile(v, 90):8.0f}"
    names = ["issue loads + barrier S1", "halo -> fp16 image (+ load wait)", "records (+ barrier)", "barrier S2", "unit loop (6 units)", "barrier S3", "exchange + barrier S4", "epilogue"]
    for role in (0, 1, 2):
        t = ts[:, role, 1:, :]            # skip the first tile (weights still arriving)
        t = t.reshape(-1, 16)
        t = t[t[:, 8] != 0]
        print(f" role {role}: {len(t)} tiles stamped; whole tile", med(t[:, 8] - t[:, 0]))
        for k, nm in enumerate(names):
            print(f"    {nm:32s}", med(t[:, k + 1] - t[:, k]))
    t = ts[:, 0, :, :]
    print(" tile-to-tile period (role 0)", med((t[:, 2, 0] - t[:, 1, 0])), " first tile start -> second tile start", med(t[:, 1, 0] - t[:, 0, 0]))
