"""Which host code issues ATen fill kernels inside the captured training step?  (torch.zeros / Tensor.zero_ / fill_ called while the stream
is capturing.)   python tools/probe/find_fills.py"""
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from centernet_amd import synth  # noqa: E402
from centernet_amd.centernet_detection import CenterNetDetection  # noqa: E402
from centernet_amd.engine import TrainStep  # noqa: E402

seen = {}


def wrap(name, fn):
    def w(*a, **k):
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            st = "".join(traceback.format_stack(limit=7)[:-1])
            seen.setdefault((name, st), 0)
            seen[(name, st)] += 1
        return fn(*a, **k)
    return w


torch.zeros = wrap("torch.zeros", torch.zeros)
torch.zeros_like = wrap("torch.zeros_like", torch.zeros_like)
torch.ones_like = wrap("torch.ones_like", torch.ones_like)
torch.Tensor.zero_ = wrap("Tensor.zero_", torch.Tensor.zero_)
torch.Tensor.fill_ = wrap("Tensor.fill_", torch.Tensor.fill_)

m = CenterNetDetection("dla_34").cuda().train()
x, tgt = synth.ctdet_batch(5, 2, 128, 128)
batch = (x.cuda(), {k: v.cuda() for k, v in tgt.items()})
step = TrainStep(m, lr=1e-4, distributed=False, graph=True)
for _ in range(4):
    step(batch)
torch.cuda.synchronize()
for (name, st), c in seen.items():
    print(f"==== {name} x{c} while capturing\n{st}")
print(f"{len(seen)} site(s)")
