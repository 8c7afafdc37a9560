"""Task heads (reference: CenterNet/models/heads.py).  Each head: 3x3 conv(+bias, ReLU fused in the GEMM
epilogue) -> 1x1 conv(+bias); outputs leave the NHWC engine as public NCHW fp32 maps [B, C, H/4, W/4]."""
import torch.nn as nn

from .. import nn as hnn
from .. import ops


class HeadConv(nn.Module):
    """heads.py:4-25."""

    def __init__(self, out_channels, intermediate_channel, head_conv):
        super().__init__()
        self.out_channels = out_channels
        self.fc = nn.Sequential(hnn.Conv2d(intermediate_channel, head_conv, 3, 1, 1, bias=True), nn.Identity(),
                                hnn.Conv2d(head_conv, out_channels, 1, 1, 0, bias=True))

    def forward(self, x):
        # the hidden ReLU's backward mask is applied in the 1x1 conv's data-gradient epilogue (no separate relu_bwd pass)
        y = self.fc[2](self.fc[0](x, relu=True, defer_relu_bwd=True), mask_dx=True)
        return ops.ToNCHWFn.apply(y, self.out_channels)

    def fill_fc_weights(self):
        for m in self.modules():
            if isinstance(m, hnn.Conv2d):
                nn.init.normal_(m.weight, std=0.001)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)


class CenterHead(nn.Module):
    """heads.py:28-50: dict of heads in declaration order; `heatmap*` heads get a -2.19 final bias."""

    def __init__(self, heads, intermediate_channel, head_conv):
        super().__init__()
        self.heads = heads
        for name, out_channel in heads.items():
            setattr(self, name, HeadConv(out_channel, intermediate_channel, head_conv))
        self.init_weights()

    def forward(self, x):
        return {name: getattr(self, name)(x) for name in self.heads.keys()}

    def init_weights(self):
        for name in self.heads.keys():
            if name.startswith("heatmap"):
                getattr(self, name).fc[-1].bias.data.fill_(-2.19)
            else:
                getattr(self, name).fill_fc_weights()
