"""Where does the bf16 error of a training-mode DLA-34 enter?  Same weights, same batch, through this package in fp32 compute mode
(the mode the reference-made goldens pin at 1e-4) and in bf16 compute mode; forward hooks compare the OUTPUT of every stage:
relative rms error (rms(bf16 - fp32) / rms(fp32)) and the worst element in units of that rms.  A stage that is a contributor of its
own shows as a jump of the rms; smooth growth ~ sqrt(depth) is the storage rounding of the bf16 activations themselves (2^-9 relative
per stored tensor).      python tools/bf16_error_growth.py [train|eval] [batch] [size]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import rng, synth
from centernet_amd.centernet_detection import CenterNetDetection

mode = sys.argv[1] if len(sys.argv) > 1 else "train"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
size = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = "cuda"
x, tgt = synth.ctdet_batch(43, B, size, size)
xg = x.to(dev)
acts = {}
for dt in (torch.float32, torch.bfloat16):
    m = CenterNetDetection("dla_34", compute_dtype=dt)
    rng.fill_state_dict(m, 43)
    m = m.to(dev).train(mode == "train")
    rec = acts.setdefault(dt, {})
    names = []
    def hook(name):
        def f(mod, inp, out):
            outs = out if isinstance(out, (list, tuple)) else [out]
            for i, o in enumerate(outs):
                if isinstance(o, torch.Tensor):
                    rec[f"{name}[{i}]" if len(outs) > 1 else name] = o.detach().float().cpu()
            if isinstance(out, dict):
                for k, o in out.items():
                    rec[f"{name}.{k}"] = o.detach().float().cpu()
        return f
    bb = m.backbone
    hs = [bb.base.base_layer.register_forward_hook(hook("base_layer"))]
    for i in range(6):
        hs.append(getattr(bb.base, f"level{i}").register_forward_hook(hook(f"level{i}")))
    for name, mod in bb.dla_up.named_children():
        hs.append(mod.register_forward_hook(hook(f"dla_up.{name}")))
        for n2, m2 in mod.named_children():
            if n2.startswith(("proj_", "node_")):
                hs.append(m2.register_forward_hook(hook(f"dla_up.{name}.{n2}")))
    for n2, m2 in bb.ida_up.named_children():
        if n2.startswith(("proj_", "node_")):
            hs.append(m2.register_forward_hook(hook(f"ida_up.{n2}")))
    for name, head in m.heads[0].named_children():
        hs.append(head.fc[0].register_forward_hook(hook(f"head.{name}.hidden(conv3x3, pre-ReLU or fused)")))
    with torch.no_grad():
        out = m(xg)[0]
    for k, v in out.items():
        rec[f"OUT {k}"] = v.detach().float().cpu()
ref, low = acts[torch.float32], acts[torch.bfloat16]
print(f"DLA-34 {mode} mode, batch {B}, {size}x{size}: bf16 compute vs fp32 compute of this package, stage outputs")
print(f"{'stage':44s} {'shape':>22s} {'rel rms':>9s} {'worst/rms':>10s} {'worst % of range':>17s}")
for k in ref:
    if k not in low or ref[k].shape != low[k].shape:
        continue
    a, b = ref[k], low[k]
    d = (b - a)
    rms = float(d.pow(2).mean().sqrt())
    sc = float(a.pow(2).mean().sqrt())
    rngv = float(a.max() - a.min())
    print(f"{k:44s} {str(tuple(a.shape)):>22s} {rms / max(sc, 1e-30):9.4f} {float(d.abs().max()) / max(rms, 1e-30):10.1f} {100 * float(d.abs().max()) / max(rngv, 1e-30):16.2f}%")

# ---- sensitivity: fp32 compute everywhere, ONE bf16 rounding (relative 2^-9 per element) injected at the output of one stage ----
print("\nfp32 compute with ONE tensor rounded to bf16 (the rounding every bf16-mode layer applies to its output), error at later stages:")
m = CenterNetDetection("dla_34", compute_dtype=torch.float32)
rng.fill_state_dict(m, 43)
m = m.to(dev).train(mode == "train")
bb = m.backbone
probe_at = ["level5", "ida_up.node_2", "OUT width_height", "OUT regression", "OUT heatmap"]
for inj in ("level2", "level3", "level4"):
    rec = {}
    def keep(name):
        def f(mod, inp, out):
            rec[name] = out.detach().float().cpu()
        return f
    def inject(mod, inp, out):
        q = out.detach().bfloat16().float()
        rec["_inj"] = float((q - out.detach()).pow(2).mean().sqrt() / out.detach().pow(2).mean().sqrt())
        out.copy_(q)
        return out
    hs = [getattr(bb.base, inj).register_forward_hook(inject), bb.base.level5.register_forward_hook(keep("level5")),
          bb.ida_up.node_2.register_forward_hook(keep("ida_up.node_2"))]
    with torch.no_grad():
        out = m(xg)[0]
    for h in hs:
        h.remove()
    for k, v in out.items():
        rec[f"OUT {k}"] = v.detach().float().cpu()
    line = f"  rounding at {inj} (injected rel rms {rec['_inj']:.5f}):"
    for k in probe_at:
        a, b = ref[k], rec[k]
        line += f"  {k} {float((b - a).pow(2).mean().sqrt() / a.pow(2).mean().sqrt()):.4f}"
    print(line)
