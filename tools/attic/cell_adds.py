"""Which consumers of shared tensors (ops.GradCell) still pay a cn_add in a DLA-34 backward pass: prints shape + the autograd
Function whose backward delivered into an occupied cell."""
import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import ops, rng, synth
from centernet_amd.centernet_detection import CenterNetDetection

orig = ops._add_tensors
def logged(a, b):
    st = traceback.extract_stack()
    who = [f.name + "@" + str(f.lineno) for f in st if f.filename.endswith("ops.py")][-4:]
    print("add", tuple(a.shape), who)
    return orig(a, b)
ops._add_tensors = logged
m = CenterNetDetection("dla_34", compute_dtype=torch.bfloat16)
rng.fill_state_dict(m, 3)
m = m.cuda().train()
x, tgt = synth.ctdet_batch(3, 2, 128, 128)
loss, _ = m.loss(m(x.cuda()), {k: v.cuda() for k, v in tgt.items()})
loss.backward()
torch.cuda.synchronize()
print("adds:", ops.GradCell.adds)
