import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import ops, rng, synth, nn as hnn
from centernet_amd.centernet_detection import CenterNetDetection

def run(seed=11):
    m = CenterNetDetection("dla_34", compute_dtype=torch.bfloat16)
    rng.fill_state_dict(m, seed)
    m = m.cuda().train()
    rec = []
    names = {mod: n for n, mod in m.named_modules()}
    def hook(mod, inp, out):
        x = inp[0]
        part = getattr(x, "_bn_part", None)
        xf = x.detach().float().reshape(-1, x.shape[-1])
        ref = torch.stack([xf.sum(0), (xf * xf).sum(0)])
        got = part.detach().clone().sum(0) if part is not None else None
        rec.append((names[mod], float(out.detach().float().abs().sum()), None if part is None else hex(part.data_ptr()),
                    None if got is None else float(((got - ref).abs() / (ref.abs().amax(1, keepdim=True) + 1e-6)).max())))
    for mod in m.modules():
        if isinstance(mod, hnn.BatchNorm2d):
            mod.register_forward_hook(hook)
    x, tgt = synth.ctdet_batch(seed, 2, 128, 128)
    loss, _ = m.loss(m(x.cuda()), {k: v.cuda() for k, v in tgt.items()})
    loss.backward(); torch.cuda.synchronize()
    return float(loss), rec
l0, r0 = run()
l1, r1 = run()
print(l0, l1)
for a, b in zip(r0, r1):
    flag = "" if abs(a[1] - b[1]) <= 1e-6 * abs(a[1]) else "   <-- differs"
    print(f"{a[0]:45s} {a[1]:14.4f} {b[1]:14.4f} sink {a[2]} / {b[2]}  sink-vs-x err {a[3]} / {b[3]}{flag}")
    if flag: break
