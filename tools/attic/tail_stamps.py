"""Where do the two streams of a REPLAYED step end?  Device wall-clock stamps (cn_stamp, 100 MHz) captured into the step's graph:
step start, end of the launch-stream chain, end of the weight-gradient stream, after the join.  No profiler attached.
    python tools/attic/tail_stamps.py [arch] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import synth, ops
from centernet_amd.centernet_detection import CenterNetDetection
from centernet_amd.decode.ctdet import ctdet_decode
from centernet_amd.engine import TrainStep

arch = sys.argv[1] if len(sys.argv) > 1 else "dla_34"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda")
m = CenterNetDetection(arch, compute_dtype=torch.bfloat16).to(dev).train()
x, t = synth.ctdet_batch(1234, bs, 512, 512)
batch = (x.to(dev), {k: v.to(dev) for k, v in t.items()})
cap = {}
l0 = m.loss
def keep(o, tg):
    r = l0(o, tg); cap["o"] = o[-1]; return r
m.loss = keep
dec = lambda: ctdet_decode(cap["o"]["heatmap"].detach(), cap["o"]["width_height"].detach(), reg=cap["o"]["regression"].detach())
ops.SideGrads.stamps = torch.zeros(4, dtype=torch.int64, device=dev)
step = TrainStep(m, lr=1e-4, graph=True, post_forward=dec, adopt_batch=True)
for _ in range(4):
    step(batch)
rows = []
for _ in range(10):
    step(batch)
    torch.cuda.synchronize()
    rows.append(ops.SideGrads.stamps.cpu().tolist())
import statistics as st
f = lambda i, j: st.median((r[j] - r[i]) / 100.0 for r in rows)        # ticks of 10 ns -> us
print(f"{arch} bs {bs}, side grid {ops.SideGrads.thin}: median over 10 replayed steps (us)")
print(f"  step start -> launch-stream chain done : {f(0, 1):9.1f}")
print(f"  step start -> weight-gradient stream done: {f(0, 2):9.1f}")
print(f"  step start -> joined                     : {f(0, 3):9.1f}")
print(f"  side stream ends {f(1, 2):+.1f} us after the launch stream (the tail in which only thin weight-gradient grids run)")
