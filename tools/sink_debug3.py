import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import ops, rng, synth
from centernet_amd.centernet_detection import CenterNetDetection

rec = []
orig_call = ops.call
def call(name, *args):
    orig_call(name, *args)
    if name == "cn_bn_train_fwd_sink":
        x, res, y, gamma, beta, rm, rv, mean, invstd, ss, part, slots, clear = args[:13]
        rec.append(("sink", x, y, mean, invstd, part.data_ptr(), None if clear is None else clear.data_ptr()))
    elif name == "cn_bn_train_fwd":
        x, res, y, gamma, beta, rm, rv, mean, invstd = args[:9]
        rec.append(("plain", x, y, mean, invstd, 0, 0))
ops.call = call

def run(seed=11):
    rec.clear()
    m = CenterNetDetection("dla_34", compute_dtype=torch.bfloat16)
    rng.fill_state_dict(m, seed)
    m = m.cuda().train()
    x, tgt = synth.ctdet_batch(seed, 2, 128, 128)
    loss, _ = m.loss(m(x.cuda()), {k: v.cuda() for k, v in tgt.items()})
    loss.backward(); torch.cuda.synchronize()
    out = []
    for kind, x_, y_, mean, invstd, p, c in rec:
        xf = x_.float().reshape(-1, x_.shape[-1])
        out.append((kind, tuple(x_.shape), mean.clone(), invstd.clone(), xf.mean(0), float(y_.float().abs().sum()), hex(p), hex(c) if c else None, float(xf.abs().sum())))
    return float(loss), out
runs = [run() for _ in range(4)]
print([r[0] for r in runs])
a = runs[0][1]
for k in range(1, 4):
    b = runs[k][1]
    for i, (u, v) in enumerate(zip(a, b)):
        dx = abs(u[8] - v[8]) / (abs(u[8]) + 1e-9)
        dm = float((u[2] - v[2]).abs().max())
        if dx > 1e-6 or dm > 1e-6:
            err_u = float((u[2] - u[4]).abs().max()); err_v = float((v[2] - v[4]).abs().max())
            print(f"run {k}: first difference at BN #{i} {u[0]} {u[1]}: input differs {dx:.2e}; saved mean differs {dm:.2e}; mean-vs-x run0 {err_u:.2e} run{k} {err_v:.2e}; sink {u[6]}/{v[6]} clear {u[7]}/{v[7]}")
            break
    else:
        print(f"run {k}: identical")
