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
rows = [a for a in prof.key_averages(group_by_input_shape=True) if a.key.startswith("aten::") and a.self_device_time_total > 0]
for a in sorted(rows, key=lambda a: -a.self_device_time_total)[:60]:
    print(f"{a.self_device_time_total:9.0f} us {a.count:4d}  {a.key:22s} {str(a.input_shapes)[:100]}")
# where they come from: python stacks of the aten ops that launch device work themselves
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step((x, t)); torch.cuda.synchronize()
rows = [a for a in prof.key_averages(group_by_stack_n=12) if a.key.startswith("aten::") and a.self_device_time_total > 0]
for a in sorted(rows, key=lambda a: -a.count)[:70]:
    st = [f.split("/")[-1] for f in a.stack if ("centernet" in f or "engine" in f or "bench" in f)]
    print(f"{a.count:4d} {a.self_device_time_total:8.0f} us {a.key:20s} " + " <- ".join(st[:4]))
