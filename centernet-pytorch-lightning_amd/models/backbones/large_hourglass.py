"""Hourglass-104 (2 stacks) on the HIP conv engine (reference: CenterNet/models/backbones/large_hourglass.py).

Same parameter names / shapes as the reference `HourglassNet` (state_dict compatible): `pre`, `kps`, `cnvs`, `inters`,
`inters_`, `cnvs_`.  An hourglass level (`kp_module`, :143-204) is `up1(x) + upsample2x(low3(low2(low1(x))))`; the
reference's pool layer is empty (:120-121) and the first `low1` residual strides by 2 instead (:325-328).  Here every
conv+BN(+skip)+ReLU runs through `nn.conv_bn_act` (one fused kernel in eval mode) and the nearest-neighbour up-sampling is
fused with the merge add (`cn_upsample2x_add`), so the up-sampled tensor never exists in HBM.
"""
import torch
import torch.nn as nn

from ... import nn as hnn
from ... import ops


class Convolution(nn.Module):
    """large_hourglass.py:11-31 — k×k conv (bias iff no BN) → BN → ReLU."""

    def __init__(self, k, cin, cout, stride=1, with_bn=True):
        super().__init__()
        self.conv = hnn.Conv2d(cin, cout, k, stride, (k - 1) // 2, bias=not with_bn)
        self.bn = hnn.BatchNorm2d(cout) if with_bn else nn.Sequential()
        self.with_bn = with_bn

    def forward(self, x):
        if self.with_bn:
            return hnn.conv_bn_act(self.conv, self.bn, x)
        return self.conv(x, relu=True)


class StemConvolution(nn.Module):
    """`convolution(7, 3, 128, stride=2)` of `pre` (:252-254) on the NCHW fp32 image."""

    def __init__(self, cout, compute_dtype):
        super().__init__()
        self.conv = hnn.StemConv(3, cout, 7, 2, 3)
        self.bn = hnn.BatchNorm2d(cout)
        self.compute_dtype = compute_dtype

    def forward(self, img):
        return hnn.stem_bn_act(self.conv, self.bn, img, self.compute_dtype)


class Residual(nn.Module):
    """large_hourglass.py:52-92 — relu(bn2(conv2(relu(bn1(conv1(x))))) + skip(x)); skip = 1×1 conv + BN when the shape changes."""

    def __init__(self, k, cin, cout, stride=1):
        super().__init__()
        self.conv1 = hnn.Conv2d(cin, cout, 3, stride, 1)
        self.bn1 = hnn.BatchNorm2d(cout)
        self.conv2 = hnn.Conv2d(cout, cout, 3, 1, 1)
        self.bn2 = hnn.BatchNorm2d(cout)
        if stride != 1 or cin != cout:
            self.skip = nn.Sequential(hnn.Conv2d(cin, cout, 1, stride), hnn.BatchNorm2d(cout))
        else:
            self.skip = nn.Sequential()

    def forward(self, x):
        if len(self.skip) == 0:
            y, idt = hnn.conv_bn_act_skip(self.conv1, self.bn1, x)
        else:
            x = ops.share(x)          # skip projection + conv1: data gradients summed in a kernel epilogue (ops.GradCell)
            idt = hnn.conv_bn_act(self.skip[0], self.skip[1], x, None, False)
            y = hnn.conv_bn_act(self.conv1, self.bn1, x)
        return hnn.conv_bn_act(self.conv2, self.bn2, y, idt, True)


def _layer(cin, cout, n, stride=1):
    """make_layer (:95-99) / make_hg_layer (:325-328): the FIRST block changes width (and strides)."""
    return nn.Sequential(Residual(3, cin, cout, stride), *[Residual(3, cout, cout) for _ in range(n - 1)])


def _layer_revr(cin, cout, n):
    """make_layer_revr (:102-107): the LAST block changes width."""
    return nn.Sequential(*[Residual(3, cin, cin) for _ in range(n - 1)], Residual(3, cin, cout))


class KpModule(nn.Module):
    """large_hourglass.py:143-204 (recursive hourglass level)."""

    def __init__(self, n, dims, modules):
        super().__init__()
        self.n = n
        cur, nxt = dims[0], dims[1]
        self.up1 = _layer(cur, cur, modules[0])
        self.max1 = nn.Sequential()
        self.low1 = _layer(cur, nxt, modules[0], stride=2)
        self.low2 = KpModule(n - 1, dims[1:], modules[1:]) if n > 1 else _layer(nxt, nxt, modules[1])
        self.low3 = _layer_revr(nxt, cur, modules[0])
        self.up2 = nn.Identity()          # nearest x2, fused into the merge below
        self.merge = nn.Identity()

    def forward(self, x):
        x = ops.share(x)              # both arms of the hourglass module read x
        up1 = self.up1(x)
        low3 = self.low3(self.low2(self.low1(x)))
        return ops.upsample2x_add(up1, low3)


class HourglassNet(nn.Module):
    """exkp (:207-322) with HourglassNet's constants (:331-348): n=5, dims [256,256,384,384,384,512], modules [2,2,2,2,2,4].
    forward(img NCHW fp32) -> [cnv_0, cnv_1] as NHWC activations (one map per stack; heads and losses are per stack)."""

    def __init__(self, num_stacks=2, compute_dtype=torch.bfloat16, n=5, dims=(256, 256, 384, 384, 384, 512),
                 modules=(2, 2, 2, 2, 2, 4), cnv_dim=256, pre_dim=128):
        super().__init__()
        self.nstack, self.out_channels, self.compute_dtype = num_stacks, cnv_dim, compute_dtype
        self.nchw_out = False                 # True: return the reference's NCHW fp32 maps instead of NHWC handles
        dims, modules = list(dims), list(modules)
        cur = dims[0]
        self.pre = nn.Sequential(StemConvolution(pre_dim, compute_dtype), Residual(3, pre_dim, cur, stride=2))
        self.kps = nn.ModuleList([KpModule(n, dims, modules) for _ in range(num_stacks)])
        self.cnvs = nn.ModuleList([Convolution(3, cur, cnv_dim) for _ in range(num_stacks)])
        self.inters = nn.ModuleList([Residual(3, cur, cur) for _ in range(num_stacks - 1)])
        self.inters_ = nn.ModuleList([nn.Sequential(hnn.Conv2d(cur, cur, 1), hnn.BatchNorm2d(cur)) for _ in range(num_stacks - 1)])
        self.cnvs_ = nn.ModuleList([nn.Sequential(hnn.Conv2d(cnv_dim, cur, 1), hnn.BatchNorm2d(cur)) for _ in range(num_stacks - 1)])

    def forward(self, img):
        inter = self.pre(img)
        outs = []
        for i in range(self.nstack):
            inter = ops.share(inter)  # the stack and (except for the last one) the inter-stack residual read it
            cnv = ops.share(self.cnvs[i](self.kps[i](inter)))      # heads + the inter-stack projection
            outs.append(cnv)
            if i < self.nstack - 1:           # relu(inters_(inter) + cnvs_(cnv)) -> residual (:312-315)
                a = hnn.conv_bn_act(self.inters_[i][0], self.inters_[i][1], inter, None, False)
                inter = hnn.conv_bn_act(self.cnvs_[i][0], self.cnvs_[i][1], cnv, a, True)
                inter = self.inters[i](inter)
        return ops.emit_maps(outs, self.out_channels, self.nchw_out)


def get_large_hourglass_net(num_layers=0, compute_dtype=torch.bfloat16, **kwargs):
    """large_hourglass.py:351-352."""
    return HourglassNet(compute_dtype=compute_dtype, **kwargs)
