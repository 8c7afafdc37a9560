"""How long does the HOST spend inside one replayed step (hipGraphLaunch of the ~700-kernel step graph + the optimizer graph)?
If the runtime enqueues a graph's kernel nodes one by one in a main-chain-first order, the side branch (weight gradients) cannot
start before the host has walked the whole launch-stream chain.   python tools/attic/replay_host_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centernet_amd import rng, synth  # noqa: E402
from centernet_amd.centernet_detection import CenterNetDetection  # noqa: E402
from centernet_amd.engine import TrainStep  # noqa: E402

m = CenterNetDetection("dla_34").cuda().train()
rng.fill_state_dict(m, 1234)
x, tgt = synth.ctdet_batch(1234, 64)
batch = (x.cuda(), {k: v.cuda() for k, v in tgt.items()})
for graph in (True, False):
    step = TrainStep(m, lr=1e-4, graph=graph, distributed=False, adopt_batch=True)
    for _ in range(4):
        step(batch)
    torch.cuda.synchronize()
    host, total = [], []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(batch)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
    print(f"graph={graph}: host time inside step() {sorted(host)[len(host) // 2]:.2f} ms, step start -> device idle {sorted(total)[len(total) // 2]:.2f} ms", flush=True)
    del step
