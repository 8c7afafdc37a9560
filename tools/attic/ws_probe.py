"""Development aid for csrc/conv3x3_ws.hip: builds the library with -DWS_PROBE into tools/_ab/, runs one head-shaped conv and prints
the median cycle count of every phase of the tile loop (wave 0 of each workgroup, tiles 1..7).
    python tools/attic/ws_probe.py build      # here (hipcc)         python tools/attic/ws_probe.py run      # on the GPU box"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "_ab", "lib_ws_probe.so")
CSRC = os.path.join(ROOT, "centernet-pytorch-lightning_amd", "csrc")

if sys.argv[1] == "build":
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    objs = [os.path.join(CSRC, "_build", f) for f in sorted(os.listdir(os.path.join(CSRC, "_build"))) if f.endswith(".o") and f != "conv3x3_ws.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DWS_PROBE", "-c",
                           os.path.join(CSRC, "conv3x3_ws.hip"), "-o", "/tmp/ws_probe.o"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, "/tmp/ws_probe.o"] + objs)
    print("built", SO)
else:
    os.environ["CN_LIB_PATH"] = SO
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from centernet_amd import _hip, ops
    Co = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    x = torch.randn(64, 128, 128, 64, device="cuda").bfloat16()
    w = torch.randn(Co, 64, 3, 3, device="cuda") * 0.04
    wp = ops.pack_weight(w, 1, torch.bfloat16)
    for _ in range(3):
        y = ops._igemm(x, wp, None, None, Co, 3, 3, 1, 1, False, True, 128, 128)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = ops._igemm(x, wp, None, None, Co, 3, 3, 1, 1, False, True, 128, 128)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    lib = ctypes.CDLL(SO)
    buf = np.zeros(512 * 8 * 8, dtype=np.uint64)
    assert lib.ws_probe_dump(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    ts = buf.reshape(512, 8, 8).astype(np.int64)          # [workgroup * 2 + wave group][tile][stamp]
    names = ["MFMAs", "vmcnt(0) + barrier 1", "epilogue", "DMA issue", "barrier 2", "loop back"]
    for grp in (0, 1):
        tg = ts[grp::2]
        d = [tg[:, 1:7, k + 1] - tg[:, 1:7, k] for k in range(5)] + [tg[:, 2:8, 0] - tg[:, 1:7, 5]]
        print("wave group", grp)
        for nm, v in zip(names, d):
            print(f"  {nm:24s} median {np.median(v):9.0f}  p10 {np.percentile(v, 10):9.0f}  p90 {np.percentile(v, 90):9.0f} cycles")
        per = float(np.median(tg[:, 2:8, 0] - tg[:, 1:7, 0]))
        ntile = 64 * 64 * max(1, Co // 64) / 256
        print(f"  tile period {per:.0f} cycles; launch {us:.1f} us / {ntile:.0f} tiles per workgroup -> shader clock ~ {per * ntile / us / 1e3:.2f} GHz")
