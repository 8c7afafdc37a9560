"""A/B of the two DCNv2 forward kernels (CN_DISABLE_DCN_FWD_TILE=1 selects the global-gather one): saves the output of a fixed
problem so two runs can be diffed, and times the launch."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centernet_amd import rng, ops
N, H, W, Ci, Co, off = [int(a) for a in sys.argv[1:6]] + [float(sys.argv[6])]
dev = "cuda"
x = rng.t_normal(1, "x", (N, H, W, Ci)).to(dev).bfloat16()
om = (rng.t_normal(1, "om", (N, H, W, 32)) * off).to(dev)
w = rng.t_normal(1, "w", (Co, Ci, 3, 3), 0, (2.0 / (Ci * 9)) ** 0.5).to(dev)
b = rng.t_normal(1, "b", (Co,), 0, 0.1).to(dev)
wp = ops.pack_weight(w, 1, torch.bfloat16, None)
y = torch.empty((N, H, W, Co), dtype=torch.bfloat16, device=dev)
def run():
    ops.call("cn_dcn_fwd", x, om, wp, b, y, N, H, W, Ci, Ci, Co, Co, 32, 1, ops.dtype_code(torch.bfloat16))
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): run()
torch.cuda.synchronize()
print("us/launch", (time.perf_counter() - t0) / 20 * 1e6)
torch.save(y.float().cpu(), sys.argv[7])
