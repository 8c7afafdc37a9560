"""Run-to-run reproducibility probe: DLA-34 bf16 gradients, several times in one process, interleaved with TrainStep graph
replays of another model (which leave BatchNorm sink / pack-arena / side-stream state behind)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import ops, rng, synth
from centernet_amd.centernet_detection import CenterNetDetection
from centernet_amd.engine import TrainStep

def grads(cells, dtype=torch.bfloat16, seed=11):
    ops.GradCell.enabled = cells
    m = CenterNetDetection("dla_34", compute_dtype=dtype)
    rng.fill_state_dict(m, seed)
    m = m.cuda().train()
    x, tgt = synth.ctdet_batch(seed, 2, 128, 128)
    loss, _ = m.loss(m(x.cuda()), {k: v.cuda() for k, v in tgt.items()})
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None}

def worst(a, b):
    w, wn = 0.0, None
    for n in a:
        d = float((a[n] - b[n]).norm()) / (float(a[n].norm()) + 1e-3 * max(float(v.norm()) for v in a.values()))
        if d > w: w, wn = d, n
    return w, wn

def other_steps(graph):
    m = CenterNetDetection("res_18", compute_dtype=torch.bfloat16)
    rng.fill_state_dict(m, 5)
    m = m.cuda().train()
    x, tgt = synth.ctdet_batch(5, 4, 128, 128)
    b = (x.cuda(), {k: v.cuda() for k, v in tgt.items()})
    st = TrainStep(m, lr=1e-4, distributed=False, graph=graph)
    for _ in range(3): st(b)
    torch.cuda.synchronize()

l0, g0 = grads(False)
for i in range(8):
    if i % 2 == 1: other_steps(graph=(i % 4 == 1))
    cells = i % 2 == 0
    l, g = grads(cells)
    print(i, "cells", cells, "loss", l, "vs first", worst(g0, g), flush=True)
