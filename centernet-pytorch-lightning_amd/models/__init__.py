"""Backbone plugin registry — same surface as CenterNet/models/__init__.py:6-19."""
import torch

from .backbones.msra_resnet import get_pose_net
from .backbones.pose_dla_dcn import get_pose_net as get_dla_dcn
from .backbones.resnet_dcn import get_pose_net as get_pose_net_dcn
from .backbones.large_hourglass import get_large_hourglass_net



_model_factory = {
    "res": get_pose_net,          # ResNet + deconv
    "dla": get_dla_dcn,           # DLA-34 + DCNv2
    "resdcn": get_pose_net_dcn,   # ResNet + DCNv2 / deconv up path
    "hourglass": get_large_hourglass_net,   # 2-stack Hourglass-104
}


def create_model(arch, compute_dtype=torch.bfloat16, **kwargs):
    """`"res_18"` -> ("res", 18) -> factory(num_layers=18).  Backbones take the NCHW fp32 image and return a list
    of NHWC activation maps with `.out_channels` channels at stride 4 (consumed by heads.CenterHead)."""
    num_layers = int(arch[arch.find("_") + 1:]) if "_" in arch else 0
    family = arch[: arch.find("_")] if "_" in arch else arch
    return _model_factory[family](num_layers=num_layers, compute_dtype=compute_dtype, **kwargs)
