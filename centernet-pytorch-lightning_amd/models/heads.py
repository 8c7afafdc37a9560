"""Task heads on the NHWC conv engine (surface of CenterNet/models/heads.py: `HeadConv`, `CenterHead`, checkpoint keys
`<head>.fc.0.*` / `<head>.fc.2.*`).

A head is conv3x3(+bias) -> ReLU -> conv1x1(+bias).  Here the ReLU lives in the first conv's GEMM epilogue and its backward
mask in the second conv's data-gradient epilogue (`defer_relu_bwd` / `mask_dx`), so a head is two forward launches and
the hidden activation is never re-read just to be masked; the result leaves the engine as a public NCHW fp32 map
[B, C, H/4, W/4] (`ops.ToNCHWFn`).

Input contract (heads.py:38-43 takes the backbone's `Tensor[B, C, H, W]`): `CenterHead.forward` accepts the reference's NCHW
fp32 tensor — converted ONCE for all heads by `ops.FromNCHWFn` (differentiable) — as well as the NHWC handle this package's
backbones emit by default (tagged by `ops.mark_nhwc`; no conversion).
"""
import torch
import torch.nn as nn

from .. import nn as hnn
from .. import ops

_NO_HEAD2 = bool(__import__("os").environ.get("CN_DISABLE_HEAD2"))      # A/B: the one-launch 2-channel head of the no-grad path
_FUSED_NODE = not __import__("os").environ.get("CN_DISABLE_HEAD_FN")      # A/B: the head as three separate autograd nodes
PRIOR_LOGIT = -2.19          # sigmoid^-1(0.1): bias of every `heatmap*` head's last conv (heads.py:45-50)


def _init_like_reference(name, head):
    """heads.py:45-50 / 19-25: heat-map heads only get the prior bias (their weights keep the framework default);
    every other head is N(0, 0.001) with zero biases."""
    hidden, out = head.fc[0], head.fc[2]
    if name.startswith("heatmap"):
        out.bias.data.fill_(PRIOR_LOGIT)
        return
    for conv in (hidden, out):
        nn.init.normal_(conv.weight, std=0.001)
        nn.init.zeros_(conv.bias)


class HeadConv(nn.Module):
    """heads.py:4-25.  `fc` keeps the reference's Sequential indices (slot 1 is where its nn.ReLU sits)."""

    def __init__(self, out_channels, intermediate_channel, head_conv):
        super().__init__()
        self.out_channels = out_channels
        hidden = hnn.Conv2d(intermediate_channel, head_conv, 3, stride=1, padding=1, bias=True)
        out = hnn.Conv2d(head_conv, out_channels, 1, stride=1, padding=0, bias=True)
        self.fc = nn.Sequential(hidden, nn.Identity(), out)

    def forward(self, x):
        hidden, out = self.fc[0], self.fc[2]
        if (_FUSED_NODE and torch.is_grad_enabled() and x.is_cuda and (x.requires_grad or hidden.weight.requires_grad)
                and hidden.k == 3 and hidden.stride == 1 and hidden.padding == 1 and hidden.bias is not None and out.bias is not None):
            # one autograd node per head: its backward works on the gathered rows when the loss is a gather-type one (ops.HeadFn)
            y = ops.HeadFn.apply(x, hidden.weight, hidden.bias, out.weight, out.bias)
            y._cn_head_dtype = x.dtype          # SigmoidFocalFn leaves a second-layout gradient only for a bf16 HeadFn (ops.DualLayout)
            return y
        if (_FUSED_NODE and not _NO_HEAD2 and self.out_channels == 2 and x.is_cuda and x.dtype == torch.bfloat16 and hidden.k == 3
                and hidden.stride == 1 and hidden.padding == 1 and hidden.bias is not None and out.bias is not None
                and not (torch.is_grad_enabled() and (x.requires_grad or hidden.weight.requires_grad or out.weight.requires_grad))):
            y = ops.head2_infer(x, hidden, out)                # no-grad 2-channel head: ONE launch, the hidden activation is never stored
            if y is not None:
                return y
        h = hidden(x, relu=True, defer_relu_bwd=True)          # ReLU in the epilogue; its backward is owed to ...
        if not (torch.is_grad_enabled() and (h.requires_grad or out.weight.requires_grad)) and _FUSED_NODE:
            return out.infer_nchw(h)                           # no-grad: the last conv writes the public NCHW fp32 map itself
        y = out(h, mask_dx=True)                               # ... this conv's data-gradient epilogue
        return ops.ToNCHWFn.apply(y, self.out_channels)

    def fill_fc_weights(self):
        _init_like_reference("", self)


class CenterHead(nn.Module):
    """heads.py:28-50: one `HeadConv` per entry of `heads` (declaration order is the output order)."""

    def __init__(self, heads, intermediate_channel, head_conv, compute_dtype=None):
        super().__init__()
        self.heads = heads
        self.intermediate_channel = intermediate_channel
        self.compute_dtype = compute_dtype      # dtype an NCHW input is converted to (None: bf16 for bf16/fp16 inputs, else fp32)
        for name, channels in heads.items():
            self.add_module(name, HeadConv(channels, intermediate_channel, head_conv))
        self.init_weights()

    def _to_engine(self, x):
        """NHWC handle -> as is; public [B, C, H, W] tensor -> NHWC activations in the compute dtype (one launch)."""
        if ops.is_nhwc(x):
            return x
        if x.dim() != 4 or x.shape[1] != self.intermediate_channel:
            raise ValueError(f"CenterHead expects [B, {self.intermediate_channel}, H, W] (or an NHWC handle of this package's "
                             f"backbones), got {tuple(x.shape)}")
        dt = self.compute_dtype or (torch.bfloat16 if x.dtype in (torch.bfloat16, torch.float16) else torch.float32)
        return ops.FromNCHWFn.apply(x, dt)

    def forward(self, x):
        x = self._to_engine(x)
        if len(self.heads) > 1:
            x = ops.share(x)      # sibling heads read one map: their data gradients are summed in the conv epilogues (ops.GradCell)
        return {name: self._modules[name](x) for name in self.heads}

    def init_weights(self):
        for name in self.heads:
            _init_like_reference(name, self._modules[name])
