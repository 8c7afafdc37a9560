import os, sys, collections
sys.path.insert(0, "/root/repo")
import torch
from centernet_amd import synth
from centernet_amd.centernet_detection import CenterNetDetection
from centernet_amd.engine import TrainStep
dev = torch.device("cuda")
m = CenterNetDetection("dla_34", compute_dtype=torch.bfloat16).to(dev).train()
x, t = synth.ctdet_batch(1, 8, 512, 512)
x = x.repeat(8, 1, 1, 1).to(dev); t = {k: v.repeat(8, *([1] * (v.dim() - 1))).to(dev) for k, v in t.items()}
step = TrainStep(m, lr=1e-4, graph=False)
for _ in range(2): step((x, t))
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    step((x, t)); torch.cuda.synchronize()
c = collections.Counter(); tm = collections.Counter()
for e in prof.events():
    if e.name in ("aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::copy_", "aten::cat", "aten::mul", "aten::sum", "aten::clone", "aten::contiguous"):
        k = (e.name, str(e.input_shapes)[:90]); c[k] += 1; tm[k] += e.device_time_total
for k, n in sorted(c.items(), key=lambda kv: -tm[kv[0]])[:40]:
    print(f"{tm[k]:9.0f} us {n:4d}  {k[0]:14s} {k[1]}")
