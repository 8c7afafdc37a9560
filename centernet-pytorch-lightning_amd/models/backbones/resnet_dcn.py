"""ResNet + deformable up path on the HIP engine (reference: CenterNet/models/backbones/resnet_dcn.py).

The trunk is msra_resnet's (same blocks, same parameter names); the up path is resnet_dcn.py:189-234 —
3 x (DCNv2 3x3 -> BN -> ReLU -> ConvTranspose2d 4x4/s2 -> BN -> ReLU) with 256/128/64 filters, `out_channels = 64`
(resnet_dcn.py:134).  Pure reuse of kernels that already exist for the DLA path: cn_dcn_fwd / cn_dcn_bwd_* for the
deformable convs, the parity-class implicit GEMM for the transposed convs, fused BN + ReLU.
"""
import math

import torch
import torch.nn as nn

from ... import nn as hnn
from ... import ops
from .msra_resnet import PoseResNet as _MsraPoseResNet, resnet_spec


def fill_up_weights(up):
    """resnet_dcn.py:108-117: bilinear kernel written into input-channel 0 of every output filter group."""
    w = up.weight.data
    f = math.ceil(w.size(2) / 2)
    c = (2 * f - 1 - f % 2) / (2.0 * f)
    k = torch.tensor([1 - abs(i / f - c) for i in range(w.size(2))], dtype=w.dtype)
    w[:, 0, :, :] = torch.outer(k, k).to(w.device)


class PoseResNet(_MsraPoseResNet):
    """resnet_dcn.py:131-261.  forward(img NCHW fp32) -> [feature map] as NHWC activations."""

    def __init__(self, block, layers, compute_dtype=torch.bfloat16, **kwargs):
        super().__init__(block, layers, compute_dtype=compute_dtype)
        self.out_channels = 64
        self.inplanes = 512 * block.expansion
        mods = []
        for planes in (256, 128, 64):                     # resnet_dcn.py:146-150
            fc = hnn.DCN(self.inplanes, planes, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1)
            up = hnn.ConvTranspose2d(planes, planes, 4, 2, 1)
            fill_up_weights(up)
            mods += [fc, hnn.BatchNorm2d(planes), nn.Identity(), up, hnn.BatchNorm2d(planes), nn.Identity()]
            self.inplanes = planes
        self.deconv_layers = nn.Sequential(*mods)

    def forward(self, img):
        x = hnn.stem_bn_act(self.conv1, self.bn1, img, self.compute_dtype)
        x = self.maxpool(x)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        d = self.deconv_layers
        for i in range(0, 18, 6):
            dcn, bn = d[i], d[i + 1]
            if not bn.training and not (torch.is_grad_enabled() and (dcn.weight.requires_grad or x.requires_grad)):
                s, b = bn.folded()
                x = dcn.infer(x, s, b, True)              # eval: BN folded into the deformable conv's GEMM
            else:
                x = bn(dcn(x), None, True)
            x = d[i + 4](d[i + 3](x))
        return ops.emit_maps([x], self.out_channels, self.nchw_out)

    def init_weights(self, num_layers, pretrained=True):
        """resnet_dcn.py:250-261 minus the ImageNet download (no network): up-path BN weights 1 / biases 0."""
        for m in self.deconv_layers.modules():
            if isinstance(m, hnn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


def get_pose_net(num_layers, compute_dtype=torch.bfloat16, pretrained_path=None):
    """resnet_dcn.py:273-278 (optional local state_dict instead of the download)."""
    block_class, layers = resnet_spec[num_layers]
    model = PoseResNet(block_class, layers, compute_dtype=compute_dtype)
    model.init_weights(num_layers)
    if pretrained_path:
        model.load_state_dict(torch.load(pretrained_path, map_location="cpu"), strict=False)
    return model
