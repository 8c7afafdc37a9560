"""DLA-34 multi_pose train step (BASELINE config 5: hm + hm_hp + wh + reg + hps heads, bs=32, 512^2) — timing only."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import synth
from centernet_amd.centernet_multi_pose import CenterNetMultiPose
from centernet_amd.engine import TrainStep
B, steps = int(os.environ.get("B", 32)), int(os.environ.get("STEPS", 15))
dev = torch.device("cuda")
m = CenterNetMultiPose("dla_34", compute_dtype=torch.bfloat16).to(dev).train()
x, t = synth.pose_batch(7, min(B, 8))
rep = (B + 7) // 8
x = x.repeat(rep, 1, 1, 1)[:B].to(dev)
t = {k: v.repeat(rep, *([1] * (v.dim() - 1)))[:B].to(dev) for k, v in t.items()}
step = TrainStep(m, lr=1e-4, graph=True)
for _ in range(3):
    loss = step((x, t))
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    loss = step((x, t))
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print(json.dumps({"metric": f"multi_pose dla_34 train step images/s (bf16, bs={B}, 512^2, hipGraph)", "value": round(B / dt, 1),
                  "ms_per_step": round(dt * 1e3, 2), "loss": round(float(loss), 4)}))
