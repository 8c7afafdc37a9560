import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import rng, synth
from oracle import models_ref, ops_ref
print("cpu_count", os.cpu_count(), "torch threads", torch.get_num_threads(), flush=True)
os.system("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Socket' ")
m = models_ref.CenterNetRef("dla_34"); rng.fill_state_dict(m, 1); m.train()
for thr in (None, 16, 32, 64):
    if thr: torch.set_num_threads(thr)
    for size, b in ((128, 2), (256, 2)):
        x, t = synth.ctdet_batch(1, b, size, size)
        t0 = time.time(); out = m(x); loss, _ = m.loss(out, t); loss.backward(); dt = time.time() - t0
        print("threads", torch.get_num_threads(), "size", size, "bs", b, f"{dt:.2f}s", flush=True)
