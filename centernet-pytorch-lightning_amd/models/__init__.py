"""Backbone plug-in point (surface of CenterNet/models/__init__.py:6-19: `_model_factory`, `create_model`).

An architecture string is `<family>[_<depth>]`; the family picks a factory, the depth is passed through as `num_layers`.
Every factory returns a module with `.out_channels` that maps the NCHW fp32 image to a list of feature maps at stride 4 (one
per stack).  `create_model(arch)` keeps the REFERENCE's return contract — `list[Tensor[B, out_channels, H/4, W/4]]`, fp32 NCHW
(models/__init__.py:14-19) — so a consumer outside this package sees exactly what the reference's backbones return.  The task
modules of this package (`centernet.CenterNet`) pass `nchw_out=False`: the maps are then the engine's NHWC activation handles
(compute dtype, tagged by `ops.mark_nhwc`) that `heads.CenterHead` consumes without a layout change.  `heads.CenterHead` accepts
either, so a plain-torch NCHW backbone registered in `_model_factory` composes with this package's heads and this package's
backbones with plain-torch heads.
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


def create_model(arch, compute_dtype=torch.bfloat16, nchw_out=True, **kwargs):
    """`"res_18"` -> family "res", num_layers 18.  Unknown families raise KeyError like the reference's dict lookup."""
    family, depth = _split_arch(arch)
    model = _model_factory[family](num_layers=depth, compute_dtype=compute_dtype, **kwargs)
    model.nchw_out = bool(nchw_out)
    return model
