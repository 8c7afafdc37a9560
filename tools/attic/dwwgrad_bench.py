"""depthwise up-conv weight gradient at the DLA-34 shapes: python tools/attic/dwwgrad_bench.py  (CN_DISABLE_DWDECONV_ROWS=1: the tap-per-lane kernel)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centernet_amd import _hip, ops
def bench(N, H, W, C):
    x = torch.randn(N, H, W, C, device="cuda").bfloat16(); dy = torch.randn(N, 2 * H, 2 * W, C, device="cuda").bfloat16()
    dw = torch.zeros(C, 1, 4, 4, device="cuda")
    fn = lambda: ops._dwdeconv_wgrad(x, dy, dw, N, H, W, C, 4, 2, 1, 2 * H, 2 * W)
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    g = torch.cuda.CUDAGraph()     # 20 launches in one hipGraph: the host's per-call overhead (ctypes, ~40 us) is off the clock
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=st):
        for _ in range(20): fn()
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / 20)
    ts.sort(); b = (x.numel() + dy.numel()) * 2
    return ts[5], b / ts[5] / 1e6
for g in (160, 1536):
    ops.SideGrads.grid = g        # cn_hooks.wgrad_blocks of the weight-gradient calls below
    print(f"grid target {g}: " + "   ".join(f"{c}: {bench(*c)[0]:.1f} us" for c in ((64, 64, 64, 64), (64, 32, 32, 128), (64, 16, 16, 256), (64, 32, 32, 64))))
