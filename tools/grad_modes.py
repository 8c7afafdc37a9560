import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from centernet_amd import rng, synth
from centernet_amd.engine import TrainStep
from centernet_amd.centernet_detection import CenterNetDetection
arch = sys.argv[1]
seed, B, size = 47, 8, 128
x, t = synth.ctdet_batch(seed, B, size, size)
batch = (x.cuda(), {k: v.cuda() for k, v in t.items()})
grads, losses = {}, {}
for tag, dt, ts in (("f32", torch.float32, True), ("f32_plain", torch.float32, False), ("bf16", torch.bfloat16, True), ("bf16_plain", torch.bfloat16, False)):
    m = CenterNetDetection(arch, compute_dtype=dt); rng.fill_state_dict(m, seed); m = m.cuda().train()
    if ts:
        step = TrainStep(m, lr=0.0, distributed=False, graph=False)
        losses[tag] = float(step(batch))
    else:
        loss, _ = m.loss(m(batch[0]), batch[1]); loss.backward(); losses[tag] = float(loss)
    torch.cuda.synchronize()
    grads[tag] = {n: p.grad.detach().double().clone() for n, p in m.named_parameters() if p.grad is not None}
print(losses)
ref = grads["f32_plain"]
for tag in ("f32", "bf16", "bf16_plain"):
    rel = {n: float((grads[tag][n] - g).norm()) / max(float(g.norm()), 1e-30) for n, g in ref.items() if float(g.norm()) > 1e-8}
    v = sorted(rel.values())
    print(tag, "median", np.median(v), "p90", v[int(0.9 * len(v))], "max", v[-1], [n for n in rel if rel[n] == v[-1]])
    names = list(rel)
    print("   first 6:", [(n[-40:], round(rel[n], 3)) for n in names[:6]], " last 4:", [(n[-30:], round(rel[n], 3)) for n in names[-4:]])
