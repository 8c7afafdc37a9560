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
post = None
if os.environ.get("DECODE"):      # multi_pose_decode of every step's head maps, forked onto its own stream next to backward
    from centernet_amd.decode.multi_pose import multi_pose_decode
    kept, orig = {}, m.loss

    def loss_and_keep(outputs, target):
        r = orig(outputs, target)
        kept["o"] = {k: v.detach() for k, v in outputs[-1].items()}
        return r
    m.loss = loss_and_keep

    def post():                  # the loss left both heat maps sigmoid-ed in place, as test_step_end's sigmoid_() does
        o = kept["o"]
        return multi_pose_decode(o["heatmap"], o["width_height"], o["keypoints"], reg=o["regression"],
                                 hm_hp=o["heatmap_keypoints"], hp_offset=o["heatmap_keypoints_offset"])
step = TrainStep(m, lr=1e-4, graph=not os.environ.get("EAGER"), post_forward=post)
for _ in range(3):
    loss = step((x, t))
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    loss = step((x, t))
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
if post is not None:
    det = step.post_out
    assert det.shape == (B, 100, 57) and bool(torch.isfinite(det).all()), det.shape
print(json.dumps({"metric": f"multi_pose dla_34 train step images/s (bf16, bs={B}, 512^2, hipGraph)", "value": round(B / dt, 1),
                  "ms_per_step": round(dt * 1e3, 2), "loss": round(float(loss), 4)}))
