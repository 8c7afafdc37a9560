"""The batched weight pack of a DLA-34 training step (the first kernel of every step): records, workgroups, bytes, time in a hipGraph.
usage: python tools/attic/pack_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centernet_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
torch.manual_seed(0)
from centernet_amd.centernet_detection import CenterNetDetection  # noqa: E402
m = CenterNetDetection("dla_34").to(DEV).train()
arena = ops.PackArena()
ops.PackArena.current, arena.recording = arena, True
x = torch.randn(2, 3, 128, 128, device=DEV)
out = m(x)
loss = sum(o.float().square().mean() for o in out[0].values())
loss.backward()
arena.build()
ops.PackArena.current = None
nrec = arena.table.shape[0]
elems = sum(v.numel() for v in arena.slots.values())
src = sum(int(r[2] * r[3] * r[4]) for r in arena.table.cpu().tolist())
print(f"records {nrec}, workgroups {arena.n_blocks}, packed elements {elems / 1e6:.1f} M ({elems * 2 / 1e6:.0f} MB bf16), source {src / 1e6:.1f} M floats ({src * 4 / 1e6:.0f} MB)")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    arena.repack()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(20):
            arena.repack()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(3):
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
print(f"cn_pack_weight_batch: {best:.1f} us  ({(src * 4 + elems * 2) / best / 1e6:.2f} TB/s)")
