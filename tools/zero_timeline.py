"""Timeline of a REPLAYED step without a profiler and without extra graph nodes: every cn_zero launch of the captured step stores
the device wall clock when it starts (cn_zero_stamps).  The launches are labelled at capture time with the stream they were
issued on (launch stream / weight-gradient stream) and the Python frame that issued them."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import synth, ops, _hip
from centernet_amd.centernet_detection import CenterNetDetection
from centernet_amd.decode.ctdet import ctdet_decode
from centernet_amd.engine import TrainStep
dev = torch.device("cuda")
m = CenterNetDetection("dla_34", compute_dtype=torch.bfloat16).to(dev).train()
x, t = synth.ctdet_batch(1234, 64, 512, 512)
batch = (x.to(dev), {k: v.to(dev) for k, v in t.items()})
cap = {}
l0 = m.loss
def keep(o, tg):
    r = l0(o, tg); cap["o"] = o[-1]; return r
m.loss = keep
dec = lambda: ctdet_decode(cap["o"]["heatmap"].detach(), cap["o"]["width_height"].detach(), reg=cap["o"]["regression"].detach())
step = TrainStep(m, lr=1e-4, graph=True, post_forward=dec, adopt_batch=True)
buf = torch.zeros(512, dtype=torch.int64, device=dev)
labels = []
orig = _hip.call
dummy = torch.zeros(4, dtype=torch.uint8, device=dev)
MARK = os.environ.get("MARK_BN", "1") == "1"      # a 4-byte cn_zero behind every BN backward: the launch stream's timeline through the backbone
def call(name, *a, **kw):
    if MARK and recording[0] and name in ("cn_bn_train_bwd_sink", "cn_bn_train_bwd", "cn_bn_train_bwd_acc"):
        r = orig(name, *a, **kw)
        labels.append(("main", 4, f"after {name} C={a[-3] if len(a) > 3 else ''}"))
        orig("cn_zero", dummy, 4)
        return r
    if name == "cn_zero" and recording[0]:
        side = ops.SideGrads.stream is not None and torch.cuda.current_stream() == ops.SideGrads.stream
        fr = [f for f in traceback.extract_stack()[:-1] if "centernet" in f.filename][-3:]
        labels.append(("side" if side else "main", a[1], " <- ".join(f"{f.name}:{f.lineno}" for f in reversed(fr))))
    return orig(name, *a, **kw)
recording = [False]
_hip.call = call; ops.call = call
import centernet_amd.engine as eng
# the capture happens inside the first call: warm-up eager steps run first (their cn_zero launches must not take slots), so register
# the buffer right before the capture by wrapping torch.cuda.graph
real_graph = torch.cuda.graph
class G(real_graph):
    def __enter__(self):
        if not labels:
            _hip.query("cn_zero_stamps", buf.data_ptr(), buf.numel()); recording[0] = True
        return super().__enter__()
    def __exit__(self, *e):
        r = super().__exit__(*e)
        recording[0] = False
        return r
torch.cuda.graph = G
step(batch)
_hip.query("cn_zero_stamps", None, 0)
n = len(labels)
for _ in range(12):
    step(batch)
torch.cuda.synchronize()
import time
t1 = time.perf_counter()
for _ in range(20):
    step(batch)
torch.cuda.synchronize()
print(f"step time with the markers: {(time.perf_counter() - t1) / 20 * 1e3:.3f} ms")
s = buf.cpu().tolist()[:n]
t0 = min(s)
print(f"{n} cn_zero launches in the captured step; times of the last replay, us from the first")
last_main = last_side = 0
rows = sorted(zip(labels, s), key=lambda kv: kv[1])
for (st, nb, where), v in rows:
    print(f"{(v - t0) / 100.0:10.1f} us  {st}  {nb:>11d} B  {where}")
lm = max(v for (st, _, _), v in rows if st == "main"); ls = max(v for (st, _, _), v in rows if st == "side")
print(f"last launch-stream marker at {(lm - t0) / 100.0:.1f} us, last weight-gradient-stream cn_zero at {(ls - t0) / 100.0:.1f} us")
