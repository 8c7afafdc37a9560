import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, traceback
from centernet_amd import ops, rng, synth
from centernet_amd.centernet_detection import CenterNetDetection

orig = ops.BnStats.acquire.__func__
count = {"n": 0}
def checked(cls, kind, C, device):
    buf = orig(cls, kind, C, device)
    torch.cuda.synchronize()
    mx = float(buf.abs().max())
    count["n"] += 1
    if mx != 0.0:
        st = cls._st()
        print(f"acquire #{count['n']} kind {kind} C {C}: DIRTY buffer handed out (max {mx:.3e}), ptr {buf.data_ptr():x}, ring size {len(cls._rings[(cls.ns, kind, int(C), str(device))])}, retired {[hex(t.data_ptr()) for t in st['retired']]}")
    return buf
ops.BnStats.acquire = classmethod(checked)

def grads(seed=11):
    m = CenterNetDetection("dla_34", compute_dtype=torch.bfloat16)
    rng.fill_state_dict(m, seed)
    m = m.cuda().train()
    x, tgt = synth.ctdet_batch(seed, 2, 128, 128)
    loss, _ = m.loss(m(x.cuda()), {k: v.cuda() for k, v in tgt.items()})
    print("forward done, loss", float(loss), "acquires", count["n"])
    loss.backward()
    torch.cuda.synchronize()
    print("backward done, acquires", count["n"])
for i in range(3):
    print("---- call", i)
    grads()
