"""Which training-mode BatchNorms of a DLA-34 step get their statistics from the producer's epilogue (cn_bn_train_fwd_stats) and which
still read their input twice (cn_bn_train_fwd), by tensor shape.   python tools/attic/bn_hook_count.py [arch]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centernet_amd import _hip, ops, synth  # noqa: E402
from centernet_amd.centernet_detection import CenterNetDetection  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "dla_34"
m = CenterNetDetection(arch, compute_dtype=torch.bfloat16).cuda().train()
x, tgt = synth.ctdet_batch(1, 4, 256, 256)
seen = collections.Counter()
last = [None]
orig = _hip.call


def spy(name, *args, **kw):
    if name in ("cn_bn_train_fwd", "cn_bn_train_fwd_stats"):
        npix, C = (args[10], args[11]) if name == "cn_bn_train_fwd" else (args[12], args[13])
        seen[(name, int(npix), int(C), last[0])] += 1
    elif name in ("cn_conv2d_fwd", "cn_conv1x1_cat_fwd", "cn_dcn_fwd", "cn_stem_conv_fwd"):
        last[0] = name
    return orig(name, *args, **kw)


_hip.call = spy
ops.call = spy
out = m(x.cuda())
loss, _ = m.loss(out, {k: v.cuda() for k, v in tgt.items()})
torch.cuda.synchronize()
tot = collections.Counter()
for (name, npix, C, prod), n in sorted(seen.items(), key=lambda kv: -kv[0][1] * kv[0][2]):
    print(f"{name:24s} {npix:8d} px x {C:4d} ch  x{n:2d}   producer {prod}")
    tot[name] += n * npix * C
print({k: f"{v / 1e6:.1f} M elements" for k, v in tot.items()})
