"""Development aid for csrc/conv3x3_kp.hip: builds the library with -DKP_PROBE into tools/_ab/, runs one conv and prints the median
cycle count of every phase of a step (waves 0 and 4 of each workgroup, steps 4..35).
    python tools/attic/kp_probe.py build      # here (hipcc)         python tools/attic/kp_probe.py run [Ci HW]     # on the GPU box"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "_ab", "lib_kp_probe.so")
CSRC = os.path.join(ROOT, "centernet-pytorch-lightning_amd", "csrc")

if sys.argv[1] == "build":
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    objs = [os.path.join(CSRC, "_build", f) for f in sorted(os.listdir(os.path.join(CSRC, "_build"))) if f.endswith(".o") and f != "conv3x3_kp.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DKP_PROBE", "-c",
                           os.path.join(CSRC, "conv3x3_kp.hip"), "-o", "/tmp/kp_probe.o"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, "/tmp/kp_probe.o"] + objs)
    print("built", SO)
else:
    os.environ["CN_LIB_PATH"] = SO
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from centernet_amd import _hip, ops
    Ci = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    HW = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    x = torch.randn(64, HW, HW, Ci, device="cuda").bfloat16()
    w = torch.randn(Ci, Ci, 3, 3, device="cuda") * (2.0 / (9 * Ci)) ** 0.5
    wp = ops.pack_weight(w, 1, torch.bfloat16)
    for _ in range(3):
        y = ops._igemm(x, wp, None, None, Ci, 3, 3, 1, 1, False, False, HW, HW)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = ops._igemm(x, wp, None, None, Ci, 3, 3, 1, 1, False, False, HW, HW)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    lib = ctypes.CDLL(SO)
    buf = np.zeros(256 * 2 * 40 * 8, dtype=np.uint64)  # [workgroup][wave group][step][stamp]
    assert lib.kp_probe_dump(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    ts = buf.reshape(256, 2, 40, 8).astype(np.int64)          # [workgroup][wave group][step][stamp]
    # non-intrusive stamps only (s_memtime returns through lgkmcnt: a stamp behind pending ds_reads would wait for them): 5 = MFMA phase
    # start (after lgkmcnt(0)), 6 = MFMAs issued, 7 = behind barrier 2
    flops = 2.0 * 64 * HW * HW * 9 * Ci * Ci
    print(f"conv3x3 {Ci}->{Ci} @{HW}^2 bs64: {us:.1f} us = {flops / us / 1e6:.0f} TF = {flops / us / 2.5e9 * 100:.1f} % of the MFMA peak")
    for grp in (0, 1):
        tg = ts[:, grp]
        ok = (tg[:, 4:36, 5] > 0) & (tg[:, 5:37, 5] > 0)
        print("wave group", grp, "(waves 0-3)" if grp == 0 else "(waves 4-7, one phase later)")
        for nm, v in (("MFMA phase: 16 MFMAs issued", tg[:, 4:36, 6] - tg[:, 4:36, 5]), ("vmcnt (waves 0-3) + barrier 2", tg[:, 4:36, 7] - tg[:, 4:36, 6]),
                      ("read phase + barrier 1 + lgkmcnt", tg[:, 5:37, 5] - tg[:, 4:36, 7]), ("step period", tg[:, 5:37, 5] - tg[:, 4:36, 5])):
            v = v[ok]
            print(f"  {nm:34s} median {np.median(v):7.0f}  p10 {np.percentile(v, 10):7.0f}  p90 {np.percentile(v, 90):7.0f} cycles")
        per = np.median((tg[:, 5:37, 5] - tg[:, 4:36, 5])[ok])
        steps = 9 * (Ci // 64) * (64 * HW * HW // 256) * max(1, Ci // 128) / 256.0
        print(f"  {steps:.0f} steps per workgroup -> shader clock ~ {per * steps / us / 1e3:.2f} GHz (ideal step: 1024 cycles)")
