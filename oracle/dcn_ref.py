"""ORACLE (test infrastructure): pure-torch DCNv2 (modulated deformable 3x3 conv).

PARITY UNPINNED: the reference imports ``DCN.dcn_v2.DCN`` (pose_dla_dcn.py:11, used at
:441-449) from the un-vendored, un-pinned pip dependency
``DCNv2 @ git+https://github.com/tteepe/DCNv2`` (requirements.txt:1).  This file restates
the published algorithm (Zhu et al., "Deformable ConvNets v2"; CharlesShang/DCNv2 lineage;
SURVEY.md Appendix A) and is anchored by known-answer tests only.

    om = conv_offset_mask(x)                 # [B, 27, H, W]; channels 0..17 offsets, 18..26 mask logits
    offset = om[:, :18]   (interleaved: channel 2k = dy of tap k, 2k+1 = dx of tap k)
    mask   = sigmoid(om[:, 18:])
    y[n,co,h,w] = bias[co] + sum_{c,k} W[co,c,k] * mask[n,k,h,w] * bilinear(x[n,c], h-1+i+dy, w-1+j+dx)
with tap k = 3*i + j and zero outside the image (each corner contributes only if in bounds).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def bilinear_zero(x, py, px):
    """x [B,C,H,W]; py,px [B,Ho,Wo] float -> [B,C,Ho,Wo]; out-of-image corners contribute 0."""
    B, C, H, W = x.shape
    h0 = torch.floor(py)
    w0 = torch.floor(px)
    lh = py - h0
    lw = px - w0
    out = 0
    flat = x.reshape(B, C, H * W)
    for dh, wh in ((0, 1 - lh), (1, lh)):
        for dw, ww in ((0, 1 - lw), (1, lw)):
            hh = h0 + dh
            wi = w0 + dw
            ok = (hh >= 0) & (hh <= H - 1) & (wi >= 0) & (wi <= W - 1)
            idx = (hh.clamp(0, H - 1) * W + wi.clamp(0, W - 1)).long()            # B,Ho,Wo
            v = torch.gather(flat, 2, idx.reshape(B, 1, -1).expand(-1, C, -1)).reshape(B, C, *py.shape[1:])
            out = out + v * (wh * ww * ok.to(x.dtype)).unsqueeze(1)
    return out


def dcn_v2_conv(x, offset, mask, weight, bias, stride=1, padding=1, dilation=1):
    """Modulated deformable conv, deformable_groups=1.  x [B,Ci,H,W], offset [B,2*kk,Ho,Wo],
    mask [B,kk,Ho,Wo] (already sigmoid), weight [Co,Ci,kh,kw]."""
    B, Ci, H, W = x.shape
    Co, _, kh, kw = weight.shape
    Ho = (H + 2 * padding - dilation * (kh - 1) - 1) // stride + 1
    Wo = (W + 2 * padding - dilation * (kw - 1) - 1) // stride + 1
    ys = (torch.arange(Ho, dtype=x.dtype) * stride - padding).view(1, Ho, 1)
    xs = (torch.arange(Wo, dtype=x.dtype) * stride - padding).view(1, 1, Wo)
    cols = []
    for i in range(kh):
        for j in range(kw):
            k = i * kw + j
            py = ys + i * dilation + offset[:, 2 * k]
            px = xs + j * dilation + offset[:, 2 * k + 1]
            cols.append(bilinear_zero(x, py, px) * mask[:, k:k + 1])
    col = torch.stack(cols, dim=2)                                             # B,Ci,kk,Ho,Wo
    y = torch.einsum("bckhw,ock->bohw", col, weight.reshape(Co, Ci, kh * kw))
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    return y


class DCN(nn.Module):
    """Drop-in for ``DCN.dcn_v2.DCN(chi, cho, kernel_size=(3,3), stride=1, padding=1, dilation=1,
    deformable_groups=1)`` with the checkpoint-visible names weight / bias / conv_offset_mask.*."""

    def __init__(self, in_channels, out_channels, kernel_size=(3, 3), stride=1, padding=1,
                 dilation=1, deformable_groups=1):
        super().__init__()
        assert deformable_groups == 1
        ks = kernel_size if isinstance(kernel_size, (tuple, list)) else (kernel_size, kernel_size)
        self.stride, self.padding, self.dilation, self.kernel_size = stride, padding, dilation, tuple(ks)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *ks))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        bound = 1.0 / math.sqrt(in_channels * ks[0] * ks[1])
        nn.init.uniform_(self.weight, -bound, bound)
        self.conv_offset_mask = nn.Conv2d(in_channels, 3 * ks[0] * ks[1], ks, stride=stride,
                                          padding=padding, bias=True)
        nn.init.zeros_(self.conv_offset_mask.weight)
        nn.init.zeros_(self.conv_offset_mask.bias)

    def forward(self, x):
        om = self.conv_offset_mask(x)
        kk = self.kernel_size[0] * self.kernel_size[1]
        offset = om[:, :2 * kk]
        mask = torch.sigmoid(om[:, 2 * kk:])
        return dcn_v2_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation)
