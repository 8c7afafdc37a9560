"""Per-parameter gradient error of the HIP path (fp32 compute) vs the torch-CPU oracle; also the oracle's own fp32-vs-fp64
noise floor, to separate summation-order noise (grows with depth through batch-stat BN) from kernel bugs."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centernet_amd import rng, synth
from oracle import models_ref

arch, size, task = sys.argv[1], int(sys.argv[2]), (sys.argv[3] if len(sys.argv) > 3 else "ctdet")
seed = 31
torch.set_num_threads(8)


def run_ref(dtype):
    ref = models_ref.CenterNetRef(arch, task=task)
    rng.fill_state_dict(ref, seed)
    ref = ref.to(dtype).train()
    x, tgt = (synth.ctdet_batch if task == "ctdet" else synth.pose_batch)(seed, 2, size, size)
    tgt = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in tgt.items()}
    out = ref(x.to(dtype))
    loss, st = ref.loss(out, tgt)
    loss.backward()
    return ref, {n: p.grad for n, p in ref.named_parameters() if p.grad is not None}, {k: float(v) for k, v in st.items()}, x, tgt


ref, g32, st32, x, tgt = run_ref(torch.float32)
_, g64, st64, _, _ = run_ref(torch.float64)
print("oracle fp32 vs fp64 losses", st32, st64)
gh = None
if torch.cuda.is_available():
    from centernet_amd.centernet_detection import CenterNetDetection
    from centernet_amd.centernet_multi_pose import CenterNetMultiPose
    m = (CenterNetDetection if task == "ctdet" else CenterNetMultiPose)(arch, compute_dtype=torch.float32)
    m.load_state_dict(ref.state_dict())
    m = m.cuda().train()
    x32, t32 = (synth.ctdet_batch if task == "ctdet" else synth.pose_batch)(seed, 2, size, size)
    loss, st = m.loss(m(x32.cuda()), {k: v.cuda() for k, v in t32.items()})
    loss.backward()
    print("hip losses", {k: float(v) for k, v in st.items()})
    gh = {n: p.grad.cpu() for n, p in m.named_parameters() if p.grad is not None}
print(f"{'param':60s} {'|g|max':>10s} {'cpu32-64':>10s} {'hip-64':>10s}")
for n in g64:
    s = float(g64[n].abs().max()) + 1e-30
    e32 = float((g32[n].double() - g64[n]).abs().max()) / s
    eh = float((gh[n].double() - g64[n]).abs().max()) / s if gh is not None and n in gh else float("nan")
    flag = " <<<" if gh is not None and eh > 20 * e32 + 1e-5 else ""
    print(f"{n:60s} {s:10.3e} {e32:10.2e} {eh:10.2e}{flag}")
