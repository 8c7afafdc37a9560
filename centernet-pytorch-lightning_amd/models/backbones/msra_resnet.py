"""ResNet + deconv up path on the HIP conv engine (reference: CenterNet/models/backbones/msra_resnet.py).

Same parameter names / shapes as the reference PoseResNet (state_dict compatible); the arithmetic is
NHWC implicit-GEMM convs (cn_conv2d_fwd), the stem is the direct 7x7 kernel, the 3 up-sampling layers are
ConvTranspose2d 4x4/s2 decomposed into 4 parity classes of 2x2 taps on the same GEMM core.
"""
import torch
import torch.nn as nn

from ... import nn as hnn
from ... import ops


class BasicBlock(nn.Module):
    """msra_resnet.py:29-58."""
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = hnn.Conv2d(cin, planes, 3, stride, 1)
        self.bn1 = hnn.BatchNorm2d(planes)
        self.conv2 = hnn.Conv2d(planes, planes, 3, 1, 1)
        self.bn2 = hnn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        if self.downsample is None:
            y, idt = hnn.conv_bn_act_skip(self.conv1, self.bn1, x)
        else:
            x = ops.share(x)          # feeds the projection AND conv1: their data gradients meet in a kernel epilogue (ops.GradCell)
            idt = hnn.conv_bn_act(self.downsample[0], self.downsample[1], x, None, False)
            y = hnn.conv_bn_act(self.conv1, self.bn1, x)
        return hnn.conv_bn_act(self.conv2, self.bn2, y, idt, True)


class Bottleneck(nn.Module):
    """msra_resnet.py:61-100."""
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = hnn.Conv2d(cin, planes, 1)
        self.bn1 = hnn.BatchNorm2d(planes)
        self.conv2 = hnn.Conv2d(planes, planes, 3, stride, 1)
        self.bn2 = hnn.BatchNorm2d(planes)
        self.conv3 = hnn.Conv2d(planes, planes * 4, 1)
        self.bn3 = hnn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        x = ops.share(x)              # identity / projection path + conv1 (ops.GradCell)
        idt = x if self.downsample is None else hnn.conv_bn_act(self.downsample[0], self.downsample[1], x, None, False)
        y = hnn.conv_bn_act(self.conv1, self.bn1, x)
        y = hnn.conv_bn_act(self.conv2, self.bn2, y)
        return hnn.conv_bn_act(self.conv3, self.bn3, y, idt, True)


resnet_spec = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]), 50: (Bottleneck, [3, 4, 6, 3]),
               101: (Bottleneck, [3, 4, 23, 3]), 152: (Bottleneck, [3, 8, 36, 3])}


class PoseResNet(nn.Module):
    """msra_resnet.py:103-207.  forward(img NCHW fp32) -> [feature map] as NHWC activations (`.nhwc = True`)."""

    def __init__(self, block, layers, compute_dtype=torch.bfloat16, **kwargs):
        super().__init__()
        self.compute_dtype = compute_dtype
        self.nchw_out = False                 # True: return the reference's NCHW fp32 maps instead of NHWC handles
        self.inplanes = 64
        self.out_channels = 256
        self.deconv_with_bias = False
        self.conv1 = hnn.StemConv(3, 64, 7, 2, 3)
        self.bn1 = hnn.BatchNorm2d(64)
        self.maxpool = hnn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        mods = []
        for _ in range(3):                                   # msra_resnet.py:120-124, 164-192
            mods += [hnn.ConvTranspose2d(self.inplanes, 256, 4, 2, 1), hnn.BatchNorm2d(256), nn.Identity()]
            self.inplanes = 256
        self.deconv_layers = nn.Sequential(*mods)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(hnn.Conv2d(self.inplanes, planes * block.expansion, 1, stride),
                                       hnn.BatchNorm2d(planes * block.expansion))
        mods = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        mods += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    def forward(self, img):
        x = hnn.stem_bn_act(self.conv1, self.bn1, img, self.compute_dtype)
        x = self.maxpool(x)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        for i in range(0, 9, 3):
            x = self.deconv_layers[i + 1](self.deconv_layers[i](x))
        return ops.emit_maps([x], self.out_channels, self.nchw_out)

    def init_weights(self, num_layers, pretrained=True):
        """msra_resnet.py:209-246 minus the ImageNet download (no network): deconv N(0, 0.001), BN 1/0."""
        for m in self.deconv_layers.modules():
            if isinstance(m, hnn.ConvTranspose2d):
                nn.init.normal_(m.weight, std=0.001)
            elif isinstance(m, hnn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


def get_pose_net(num_layers, compute_dtype=torch.bfloat16, pretrained_path=None):
    """msra_resnet.py:258-263.  The reference downloads ImageNet weights here; offline we take an optional
    local state_dict file instead."""
    block_class, layers = resnet_spec[num_layers]
    model = PoseResNet(block_class, layers, compute_dtype=compute_dtype)
    model.init_weights(num_layers)
    if pretrained_path:
        model.load_state_dict(torch.load(pretrained_path, map_location="cpu"), strict=False)
    return model
