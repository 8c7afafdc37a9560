"""GPU debug: one ResNet BasicBlock (train-mode BN) and big-M convs vs torch CPU, printing per-tensor errors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from centernet_amd import rng, ops
from centernet_amd.models.backbones import msra_resnet as R
from oracle import models_ref as O

def nhwc(x): return x.detach().permute(0,2,3,1).contiguous().cuda()
def nchw(y): return y.detach().float().cpu().permute(0,3,1,2)
def err(a,b): return float((a-b).abs().max())/max(1e-12,float(b.abs().max()))

for (N,C,H) in [(2,128,32),(2,64,64),(2,256,16),(2,128,16),(4,128,32)]:
    x = rng.t_normal(1,"x",(N,C,H,H)); w = rng.t_normal(1,"w",(C,C,3,3),0,(2/(9*C))**.5)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr,wr,None,1,1); gy = rng.t_normal(1,"g",tuple(yr.shape)); yr.backward(gy)
    xg, wg = nhwc(x).requires_grad_(True), w.cuda().requires_grad_(True)
    y = ops.conv2d(xg,wg,None,1,1,False); y.backward(nhwc(gy))
    print(f"conv3x3 N{N} C{C} H{H}: fwd {err(nchw(y),yr):.2e} dgrad {err(nchw(xg.grad),xr.grad):.2e} wgrad {err(wg.grad.cpu(),wr.grad):.2e}")
    # per-row error map of dgrad
    d = (nchw(xg.grad)-xr.grad).abs().amax(dim=(1,3))
    print("   dgrad err by (n,row):", [f"{v:.1e}" for v in d.flatten().tolist()[:40]])

    blk_r = O.ResBasic(C,C); rng.fill_state_dict(blk_r, 3); blk_r.train()
    blk = R.BasicBlock(C,C); blk.load_state_dict(blk_r.state_dict()); blk.cuda().train()
    xr = x.clone().requires_grad_(True); yr = blk_r(xr); yr.backward(gy)
    xg = nhwc(x).requires_grad_(True); y = blk(xg); y.backward(nhwc(gy))
    print(f"  block: fwd {err(nchw(y),yr):.2e} dx {err(nchw(xg.grad),xr.grad):.2e}", {n: f"{err(p.grad.cpu(), dict(blk_r.named_parameters())[n].grad):.1e}" for n,p in blk.named_parameters()})
