import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from centernet_amd import synth, _hip
from centernet_amd.centernet_detection import CenterNetDetection
from centernet_amd.utils.decode import sigmoid_clamped
dev = torch.device("cuda")
model = CenterNetDetection("dla_34", compute_dtype=torch.bfloat16).to(dev).eval()
x, _ = synth.ctdet_batch(1234, 8, 512, 512)
x = x.repeat(8, 1, 1, 1).to(dev)
with torch.no_grad():
    out = model(x)[-1]
    heat = sigmoid_clamped(out["heatmap"])
B, C, H, W = heat.shape
m = heat[0, 0]
print("map0 unique values", m.unique().numel(), "min", m.min().item(), "max", m.max().item())
pooled = torch.nn.functional.max_pool2d(heat, 3, 1, 1)
keep = (pooled == heat)
print("kept fraction", keep.float().mean().item())
u = [heat[b, c].unique().numel() for b in range(2) for c in range(0, 80, 10)]
print("unique per map", u)
kk = (heat * keep)[0, 0]
vals, cnt = kk[kk > 0].unique(return_counts=True)
print("kept values top", vals[-5:].tolist(), cnt[-5:].tolist())
s = torch.empty(B, C, 100, device=dev); i = torch.empty(B, C, 100, dtype=torch.int32, device=dev)
for _ in range(3): _hip.call("cn_topk_channel", heat, s, i, B, C, H, W, 100, 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): _hip.call("cn_topk_channel", heat, s, i, B, C, H, W, 100, 1)
e1.record(); torch.cuda.synchronize()
print("topk us", e0.elapsed_time(e1) * 100)
for val in (0.25, 0.10087862610816956, 0.1, 0.5):
    h2 = torch.full((B, C, H, W), val, device=dev)
    for _ in range(3): _hip.call("cn_topk_channel", h2, s, i, B, C, H, W, 100, 1)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10): _hip.call("cn_topk_channel", h2, s, i, B, C, H, W, 100, 1)
    e1.record(); torch.cuda.synchronize()
    print("flat", val, "topk us", e0.elapsed_time(e1) * 100, "first inds", i[0, 0, :4].tolist())
