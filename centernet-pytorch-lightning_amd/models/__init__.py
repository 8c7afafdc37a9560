"""Backbone plug-in point (surface of CenterNet/models/__init__.py:6-19: `_model_factory`, `create_model`).

An architecture string is `<family>[_<depth>]`; the family picks a factory, the depth is passed through as `num_layers`.
Every factory returns a module that maps the NCHW fp32 image to a list of NHWC activation maps (one per stack) with
`.out_channels` channels at stride 4 — what `heads.CenterHead` consumes.
"""
import torch

from .backbones import large_hourglass, msra_resnet, pose_dla_dcn, resnet_dcn

_model_factory = {}


def _register(family, factory):
    _model_factory[family] = factory


_register("res", msra_resnet.get_pose_net)                          # ResNet trunk + 3 full deconvs
_register("resdcn", resnet_dcn.get_pose_net)                        # ResNet trunk + (DCNv2, depthwise-free deconv) x 3
_register("dla", pose_dla_dcn.get_pose_net)                         # DLA-34 + DCNv2 up-path
_register("hourglass", large_hourglass.get_large_hourglass_net)     # 2-stack Hourglass-104


def _split_arch(arch):
    family, _, depth = arch.partition("_")
    return family, (int(depth) if depth else 0)


def create_model(arch, compute_dtype=torch.bfloat16, **kwargs):
    """`"res_18"` -> family "res", num_layers 18.  Unknown families raise KeyError like the reference's dict lookup."""
    family, depth = _split_arch(arch)
    return _model_factory[family](num_layers=depth, compute_dtype=compute_dtype, **kwargs)
