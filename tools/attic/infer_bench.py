"""Inference throughput (eval forward + ctdet_decode), hipGraph replay.  Not the bench.py metric: reported in DESIGN.md
next to BASELINE.md's inference target."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="dla_34"); ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--size", type=int, default=512); ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--task", default="ctdet")
ap.add_argument("--two-step", action="store_true", help="sigmoid_clamped pass + ctdet_decode instead of the decode on logits")
args = ap.parse_args()
from centernet_amd import synth
from centernet_amd.centernet_detection import CenterNetDetection
from centernet_amd.decode.ctdet import ctdet_decode
from centernet_amd.utils.decode import sigmoid_clamped
dev = torch.device("cuda")
model = CenterNetDetection(args.arch, compute_dtype=torch.bfloat16).to(dev).eval()
x, _ = synth.ctdet_batch(1234, min(args.batch, 8), args.size, args.size)
x = x.repeat((args.batch + 7) // 8, 1, 1, 1)[:args.batch].to(dev)

def infer():
    with torch.no_grad():
        out = model(x)[-1]
        if args.two_step:
            return ctdet_decode(sigmoid_clamped(out["heatmap"]), out["width_height"], reg=out["regression"])
        return ctdet_decode(out["heatmap"], out["width_height"], reg=out["regression"], logits_clamp=1e-4)

for _ in range(3):
    det = infer()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    det = infer()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(args.steps):
    g.replay()
torch.cuda.synchronize()
dt = (time.time() - t0) / args.steps
print(json.dumps({"metric": f"inference images/s ({args.arch} eval forward + ctdet_decode, bf16, bs={args.batch}, {args.size}^2, hipGraph)",
                  "value": round(args.batch / dt, 1), "ms_per_batch": round(dt * 1e3, 3), "det_shape": list(det.shape)}))
