"""Race screen for conv3x3_ws_kernel / conv1x1_stream_kernel: the same launch repeated under memory contention from a second
stream must give bit-identical results every time (the kernels are deterministic; an LDS-DMA read that races its data shows up as
a rare differing tile).  python tools/attic/ws_stress.py [iterations]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centernet_amd import ops  # noqa: E402

DEV = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dt = torch.bfloat16
g = torch.Generator(device="cpu").manual_seed(11)
cases = []
for name, (N, HW, Ci, Co, k) in {"ws 64->256": (64, 128, 64, 256, 3), "ws 64->64": (64, 128, 64, 64, 3), "ws 64->27": (64, 128, 64, 27, 3),
                                  "1x1 256->80": (64, 128, 256, 80, 1), "1x1 128->128": (64, 64, 128, 128, 1)}.items():
    x = torch.randn(N, HW, HW, Ci, generator=g).to(dt).to(DEV)
    w = (torch.randn(Co, Ci, k, k, generator=g) * (2.0 / (k * k * Ci)) ** 0.5).to(DEV)
    cases.append((name, x, ops.pack_weight(w, 1, dt), torch.randn(Co, generator=g).to(DEV), Co, k, HW))
noise_a = torch.randn(64 * 1024 * 1024, device=DEV)
noise_b = torch.empty_like(noise_a)
side = torch.cuda.Stream()
bad = 0
for name, x, wp, b, Co, k, HW in cases:
    ref = ops._igemm(x, wp, b, None, Co, k, k, 1, k // 2, False, True, HW, HW)
    torch.cuda.synchronize()
    refsum = ref.view(torch.int16).to(torch.int64).sum().item()
    nbad = 0
    for i in range(iters):
        if i % 2 == 0:
            with torch.cuda.stream(side):
                noise_b.copy_(noise_a)                      # 512 MB of HBM traffic next to the conv
        y = ops._igemm(x, wp, b, None, Co, k, k, 1, k // 2, False, True, HW, HW)
        if y.view(torch.int16).to(torch.int64).sum().item() != refsum or not torch.equal(y, ref):
            nbad += 1
    torch.cuda.synchronize()
    print(f"{name:14s} {iters} launches, {nbad} differ from the first", flush=True)
    bad += nbad
sys.exit(1 if bad else 0)
