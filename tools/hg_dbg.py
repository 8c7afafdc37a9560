"""Hourglass train-mode fixture: error of the HIP fp32 path against the golden (reference fp32) per stack.
Noise floor measured in the build container with the oracle in fp64 against the same golden: feat 4e-6 / 5e-5 (stack 0 / 1),
width_height 7e-6 / 1.2e-4, regression 9e-6 / 8e-5, heatmap 1e-6 / 1e-6.  HIP fp32 (MI355X): 1e-5 / 1.8e-4, 2.4e-5 / 4e-4, 2.7e-5 / 3.3e-4."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from centernet_amd import rng, synth
from centernet_amd.centernet_detection import CenterNetDetection
from conftest import strided
g = np.load("/root/repo/tests/golden/hourglass_train.npz")
m = CenterNetDetection("hourglass", compute_dtype=torch.float32)
rng.fill_state_dict(m, int(g["seed"]), var_scale=float(g["var_scale"]))
m = m.cuda().train()
x, tgt = synth.ctdet_batch(int(g["seed"]), 2, 256, 256)
feats = m.backbone(x.cuda())
for s_, f in enumerate(feats):
    from centernet_amd import ops
    fn = f.permute(0, 3, 1, 2).float()
    r = g[f"feat{s_}_s"]; got = strided(fn).cpu().numpy()
    print("feat", s_, np.abs(got - r).max() / np.abs(r).max(), np.abs(r).max())
outs = m(x.cuda())
for s_, o in enumerate(outs):
    for k in ("heatmap", "width_height", "regression"):
        r = g[f"{k}{s_}_s"]; got = strided(o[k]).detach().cpu().numpy()
        print(k, s_, np.abs(got - r).max() / np.abs(r).max())
tg = {k: v.cuda() for k, v in tgt.items()}
loss, st = m.loss(outs, tg)
loss.backward()
params = dict(m.named_parameters())
for key in g.files:
    if key.startswith("g:") and key.endswith(":s"):
        n = key[2:-2]; ref = g[key].astype(np.float64); r64 = g["g64:" + n + ":s"]
        got = strided(params[n].grad, 512).cpu().numpy().astype(np.float64)
        print(n, "mine->64 %.3e  ref32->64 %.3e  mine->ref32 %.3e" % (np.linalg.norm(got - r64) / np.linalg.norm(r64),
              np.linalg.norm(ref - r64) / np.linalg.norm(r64), np.linalg.norm(got - ref) / np.linalg.norm(ref)))
