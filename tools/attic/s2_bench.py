"""stride-2 3x3 forward convs and their data gradients at the DLA-34 shapes (hipGraph of 10 launches each): python tools/attic/s2_bench.py
(CN_DISABLE_CONV3X3_S2=1 / CN_DISABLE_DGRAD3X3_S2=1: implicit GEMM)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centernet_amd import ops
out = []
for HW, Ci, Co in ((256, 32, 64), (128, 64, 128), (64, 128, 256), (32, 256, 512)):
    x = torch.randn(64, HW, HW, Ci, device="cuda").bfloat16()
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * (2.0 / (9 * Ci)) ** 0.5
    wp = ops.pack_weight(w, 1, torch.bfloat16)
    fn = lambda: ops._igemm(x, wp, None, None, Co, 3, 3, 2, 1, False, False, HW // 2, HW // 2)
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for _ in range(10): fn()
    ts = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 100)
    ts.sort()
    fl = 2.0 * 64 * (HW // 2) ** 2 * 9 * Ci * Co
    by = 64 * HW * HW * Ci * 2 + 64 * (HW // 2) ** 2 * Co * 2
    out.append(f"{Ci}->{Co}@{HW}: {ts[4]:6.1f} us {fl / ts[4] / 2.5e9 * 100:4.1f}% {by / ts[4] / 1e6:4.2f}TB/s")
print("fwd    " + "   ".join(out))
out = []
for HW, Ci, Co in ((256, 32, 64), (128, 64, 128), (64, 128, 256), (32, 256, 512)):
    dy = torch.randn(64, HW // 2, HW // 2, Co, device="cuda").bfloat16()
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * (2.0 / (9 * Ci)) ** 0.5
    wpd = ops.pack_weight(w, 0, torch.bfloat16)
    fn = lambda: ops._igemm(dy, wpd, None, None, Ci, 3, 3, 2, 1, True, False, HW, HW)
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for _ in range(10): fn()
    ts = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 100)
    ts.sort()
    fl = 2.0 * 64 * (HW // 2) ** 2 * 9 * Ci * Co
    by = 64 * HW * HW * Ci * 2 + 64 * (HW // 2) ** 2 * Co * 2
    out.append(f"{Co}->{Ci}@{HW // 2}->{HW}: {ts[4]:6.1f} us {fl / ts[4] / 2.5e9 * 100:4.1f}% {by / ts[4] / 1e6:4.2f}TB/s")
print("dgrad  " + "   ".join(out))
