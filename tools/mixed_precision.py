"""Round-4 VERDICT item 5: what does "fp32 storage for the early DLA levels, bf16 above" buy and cost?  Training-mode DLA-34 ctdet,
same weights and batch: (a) error of the head maps against this package's fp32 mode (the mode pinned at 1e-4 to the reference goldens)
in the metric of tests/test_gpu_configs.py (_rel_range_err: worst element and rms relative to the reference's range), (b) ms per train
step (eager, 512x512, batch 64 unless given).      python tools/mixed_precision.py [batch] [size]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centernet_amd import rng, synth  # noqa: E402
from centernet_amd.centernet_detection import CenterNetDetection  # noqa: E402
from centernet_amd.engine import TrainStep  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
x, tgt = synth.ctdet_batch(1234, B, size, size)
batch = (x.cuda(), {k: v.cuda() for k, v in tgt.items()})


def rel_range_err(got, ref):
    d = (got - ref).abs()
    rng_ = (ref.max() - ref.min()).clamp_min(1e-12)
    return float(d.max() / rng_), float(d.pow(2).mean().sqrt() / rng_)


def maps(dt, fp32_levels):
    m = CenterNetDetection("dla_34", compute_dtype=dt)
    rng.fill_state_dict(m, 1234)
    m = m.cuda().train()
    m.backbone.base.fp32_levels = fp32_levels
    with torch.no_grad():
        out = m(batch[0])[0]
    res = {k: v.detach().float().cpu() for k, v in out.items()}
    # timing: eager train steps (the graph replay hides launch gaps, not kernel time; differences between the modes are kernel time)
    step = TrainStep(m, lr=1e-4, graph=False, distributed=False)
    for _ in range(2):
        step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        step(batch)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 4 * 1e3
    del step, m
    torch.cuda.empty_cache()
    return res, ms


ref, ms32 = maps(torch.float32, 0)
print(f"DLA-34 ctdet train mode, batch {B}, {size}x{size}; reference = fp32 compute mode ({ms32:.1f} ms per eager step)")
print(f"{'mode':34s} {'ms/step':>8s}  " + "  ".join(f"{k + ' worst / rms':>30s}" for k in ref))
for lv in (0, 1, 2, 3, 4):
    got, ms = maps(torch.bfloat16, lv)
    cells = []
    for k in ref:
        mx, rms = rel_range_err(got[k], ref[k])
        cells.append(f"{mx * 100:12.2f} % / {rms * 100:6.3f} %")
    name = "bf16 throughout" if lv == 0 else f"fp32 base_layer..level{lv - 1}, bf16 above"
    print(f"{name:34s} {ms:8.2f}  " + "  ".join(f"{c:>30s}" for c in cells), flush=True)
