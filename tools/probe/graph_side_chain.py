"""Replayed hipGraph: main chain of N spin kernels on stream A; after main node i a side closure of M short kernels on stream B that
depends on node i (event) and, by stream order, on the previous closure — the shape of TrainStep's weight-gradient stream.  When
does closure j actually start?  stamps: one per closure (its first kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from centernet_amd import _hip
dev = torch.device("cuda")
N = int(os.environ.get("N", 200)); M = int(os.environ.get("M", 3))
spin, sspin = 100000, 20000
stamps = torch.zeros(2 * N + 8, dtype=torch.int64, device=dev)
BIG = int(os.environ.get("BIG", 0))          # main kernels = cn_zero over BIG MB (thousands of workgroups: the chip is full) instead of a 1-thread spin
big = torch.empty(max(BIG, 1) << 20, dtype=torch.uint8, device=dev)
SIDEBIG = int(os.environ.get("SIDEBIG", 0))  # side kernels = cn_zero over SIDEBIG MB
sbig = torch.empty(max(SIDEBIG, 1) << 20, dtype=torch.uint8, device=dev)
def main_kernel():
    if BIG: _hip.call("cn_zero", big, big.numel())
    else: torch.cuda._sleep(spin)
def side_kernel():
    if SIDEBIG: _hip.call("cn_zero", sbig, sbig.numel())
    else: torch.cuda._sleep(sspin)
st = lambda i: _hip.call("cn_stamp", stamps[i:])
A, B = torch.cuda.Stream(), torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
deferred = []
def flush():
    while deferred:
        ev, j = deferred.pop(0)
        B.wait_event(ev)
        with torch.cuda.stream(B):
            st(1 + 2 * j)
            for _ in range(M):
                side_kernel()
with torch.cuda.graph(g, stream=A):
    st(0)
    for i in range(N):
        main_kernel()
        st(2 + 2 * i)                      # main node i done
        ev = torch.cuda.Event(); ev.record(A)
        if os.environ.get("DEFER", "1") == "1":
            flush()                        # like SideGrads.submit: the previous closure is launched when the next one is submitted
            deferred.append((ev, i))
        else:
            deferred.append((ev, i)); flush()
    flush()
    A.wait_stream(B)
    st(2 * N + 2)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
s = stamps.cpu().tolist()
us = lambda i: (s[i] - s[0]) / 100.0
print(f"BIG={BIG} SIDEBIG={SIDEBIG} N={N} M={M} DEFER={os.environ.get('DEFER', '1')}: main end {us(2 * N):.0f} us, joined {us(2 * N + 2):.0f} us")
for j in (0, 1, 2, 5, 10, 20, 50, 100, 150, N - 1):
    if j < N:
        print(f"  closure {j:3d}: main node done at {us(2 + 2 * j):8.1f} us, closure starts at {us(1 + 2 * j):8.1f} us  (lag {us(1 + 2 * j) - us(2 + 2 * j):8.1f})")
