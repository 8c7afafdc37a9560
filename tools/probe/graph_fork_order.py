"""How does a replayed hipGraph schedule a forked branch?  Main chain of N spin kernels (~100 us each) on stream A; ONE side kernel
(a wall-clock stamp) on stream B that depends on main node #k only.  Variants: the side kernel is CAPTURED right after node k
(interleaved capture) or after all N main nodes (late capture).  Prints when the side stamp runs relative to the chain's start."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from centernet_amd import _hip
dev = torch.device("cuda")
N, K = 60, 3
spin = 200000
stamps = torch.zeros(8, dtype=torch.int64, device=dev)

def stamp(i):
    _hip.call("cn_stamp", stamps[i:])

def run(variant, nside=1, tail_nodes=0):
    A, B = torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=A):
        stamp(0)
        evs = []
        for i in range(N):
            torch.cuda._sleep(spin)
            if i == K:
                ev = torch.cuda.Event(); ev.record(A)
                if variant == "interleaved":
                    B.wait_event(ev)
                    with torch.cuda.stream(B):
                        for j in range(nside):
                            torch.cuda._sleep(spin // 4)
                        stamp(1)
        if variant == "late":
            B.wait_event(ev)
            with torch.cuda.stream(B):
                for j in range(nside):
                    torch.cuda._sleep(spin // 4)
                stamp(1)
        stamp(2)
        with torch.cuda.stream(B):
            for j in range(tail_nodes):
                torch.cuda._sleep(spin // 4)
            stamp(4)
        A.wait_stream(B)
        stamp(3)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    s = stamps.cpu().tolist()
    us = lambda i: (s[i] - s[0]) / 100.0
    print(f"{variant:12s} nside={nside} tail={tail_nodes}: side stamp at {us(1):9.1f} us, main end {us(2):9.1f} us, side end {us(4):9.1f}, joined {us(3):9.1f} us   (node {K} ends at ~{(K + 1) * us(2) / N:.0f} us)")

for v in ("interleaved", "late"):
    run(v)
    run(v, nside=8)
    run(v, nside=8, tail_nodes=4)
