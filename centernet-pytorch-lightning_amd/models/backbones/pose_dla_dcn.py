"""DLA-34 + DCNv2 up path on the HIP engine (reference: CenterNet/models/backbones/pose_dla_dcn.py).

State-dict compatible with the reference DLASeg("dla34", down_ratio=4, last_level=5): 386 entries, e.g.
`base.level2.tree1.conv1.weight`, `dla_up.ida_0.proj_1.conv.conv_offset_mask.weight`, `ida_up.up_2.weight`.
Dataflow notes that matter for parity:
  * the outer (levels=2) Trees of level3/level4 compute `project(bottom)` and throw it away
    (pose_dla_dcn.py:252-258) — here it is only run (without autograd) in training mode so the BN running
    statistics of those dead branches evolve exactly as in the reference; it costs 0.2 % of the FLOPs;
  * IDAUp mutates its `layers` list in place (:482-488); DLASeg clones the first three maps (:567).
"""
import math

import os

import torch
import torch.nn as nn

from ... import nn as hnn
from ... import ops


class BasicBlock(nn.Module):
    """pose_dla_dcn.py:28-68 — the residual joins before the last ReLU."""

    def __init__(self, inplanes, planes, stride=1, dilation=1):
        super().__init__()
        assert dilation == 1
        self.conv1 = hnn.Conv2d(inplanes, planes, 3, stride, 1)
        self.bn1 = hnn.BatchNorm2d(planes)
        self.conv2 = hnn.Conv2d(planes, planes, 3, 1, 1)
        self.bn2 = hnn.BatchNorm2d(planes)
        self.stride = stride

    def forward(self, x, residual=None):
        if residual is None:            # identity block: the skip gradient joins conv1's data gradient in that kernel's epilogue
            y, residual = hnn.conv_bn_act_skip(self.conv1, self.bn1, x)
        else:
            y = hnn.conv_bn_act(self.conv1, self.bn1, x)
        return hnn.conv_bn_act(self.conv2, self.bn2, y, residual, True)


class Root(nn.Module):
    """pose_dla_dcn.py:165-188: 1x1 conv over the channel concatenation of the children + BN (+children[0]) + ReLU."""

    def __init__(self, in_channels, out_channels, kernel_size, residual):
        super().__init__()
        assert kernel_size == 1
        self.conv = hnn.Conv2d(in_channels, out_channels, 1)
        self.bn = hnn.BatchNorm2d(out_channels)
        self.residual = residual

    def forward(self, *xs):
        return hnn.cat_conv_bn_act(self.conv, self.bn, list(xs), xs[0] if self.residual else None, True)


_POOL_TWICE = bool(os.environ.get("CN_DLA_POOL_TWICE"))      # A/B switch: pool a two-level tree's input twice, as the reference does


class Tree(nn.Module):
    """pose_dla_dcn.py:191-265."""

    def __init__(self, levels, block, in_channels, out_channels, stride=1, level_root=False, root_dim=0,
                 root_kernel_size=1, dilation=1, root_residual=False):
        super().__init__()
        if root_dim == 0:
            root_dim = 2 * out_channels
        if level_root:
            root_dim += in_channels
        if levels == 1:
            self.tree1 = block(in_channels, out_channels, stride, dilation=dilation)
            self.tree2 = block(out_channels, out_channels, 1, dilation=dilation)
            self.root = Root(root_dim, out_channels, root_kernel_size, root_residual)
        else:
            self.tree1 = Tree(levels - 1, block, in_channels, out_channels, stride, root_dim=0,
                              root_kernel_size=root_kernel_size, dilation=dilation, root_residual=root_residual)
            self.tree2 = Tree(levels - 1, block, out_channels, out_channels, root_dim=root_dim + out_channels,
                              root_kernel_size=root_kernel_size, dilation=dilation, root_residual=root_residual)
        self.level_root, self.root_dim, self.levels = level_root, root_dim, levels
        self.downsample = hnn.MaxPool2d(stride, stride) if stride > 1 else None
        self.project = None
        if in_channels != out_channels:
            self.project = nn.Sequential(hnn.Conv2d(in_channels, out_channels, 1), hnn.BatchNorm2d(out_channels))

    def forward(self, x, residual=None, children=None, bottom=None):
        """`bottom`: the caller's own `downsample(x)` when it is the same pooling of the same tensor (an outer tree and its
        `tree1` are built with the same stride: the reference pools x twice, pose_dla_dcn.py:245-262; sharing the result drops one
        max-pool forward/backward pair and one full-resolution gradient accumulation per two-level tree — same values)."""
        children = [] if children is None else children
        if bottom is None:
            # x feeds the pooling AND tree1; the pooled tensor feeds project / the inner tree / a Root: shared tensors sum their
            # consumers' gradients in those consumers' own epilogues (ops.GradCell) instead of in autograd's add passes
            x = ops.share(x)
            bottom = ops.share(self.downsample(x)) if self.downsample is not None else x
        if self.levels == 1:
            residual = hnn.conv_bn_act(self.project[0], self.project[1], bottom, None, False) if self.project else bottom
        elif self.project is not None and self.project[1].training:
            with torch.no_grad():   # dead branch of the reference: only its BN running statistics are observable
                hnn.conv_bn_act(self.project[0], self.project[1], bottom.detach(), None, False)
        if self.level_root:
            children.append(bottom)
        if self.levels == 1:
            # stride-1, same-width tree: the residual IS x, so let the block take its own skip path (the skip gradient then
            # joins conv1's data gradient in that kernel's epilogue instead of in an element-wise pass of the autograd engine)
            x1 = ops.share(self.tree1(x, None if residual is x else residual))
            return self.root(self.tree2(x1), x1, *children)
        x1 = ops.share(self.tree1(x, bottom=bottom if self.downsample is not None and not _POOL_TWICE else None))
        children.append(x1)
        return self.tree2(x1, children=children)


class DLA(nn.Module):
    """pose_dla_dcn.py:268-378.  forward(img NCHW fp32) -> 6 NHWC maps (strides 1..32)."""

    def __init__(self, levels, channels, block=BasicBlock, compute_dtype=torch.bfloat16):
        super().__init__()
        self.channels = channels
        self.compute_dtype = compute_dtype
        # forward() returns all six levels like the reference (pose_dla_dcn.py:372-378), so level0's output is materialised
        # (BN + ReLU applied) by default.  A caller that never reads y[0] (DLASeg: first_level >= 1) clears this, and level0's last BN
        # is then left to level1's first conv (applied on load; y[0] is None instead of a raw, un-normalised tensor).
        self.expose_level0 = True
        # mixed precision (round-4 VERDICT item 5, measured in profiles/r05_mixed_precision.txt): the first `fp32_levels` stages
        # (1 = base_layer + level0, 2 = + level1, 3 = + level2, ...) compute and store fp32, everything behind them runs in
        # `compute_dtype`.  0 (default) = one dtype throughout.
        self.fp32_levels = int(os.environ.get("CN_DLA_FP32_LEVELS", 0))
        self.base_layer = nn.Sequential(hnn.StemConv(3, channels[0], 7, 1, 3), hnn.BatchNorm2d(channels[0]), nn.Identity())
        self.level0 = self._make_conv_level(channels[0], channels[0], levels[0])
        self.level1 = self._make_conv_level(channels[0], channels[1], levels[1], stride=2)
        self.level2 = Tree(levels[2], block, channels[1], channels[2], 2, level_root=False)
        self.level3 = Tree(levels[3], block, channels[2], channels[3], 2, level_root=True)
        self.level4 = Tree(levels[4], block, channels[3], channels[4], 2, level_root=True)
        self.level5 = Tree(levels[5], block, channels[4], channels[5], 2, level_root=True)

    @staticmethod
    def _make_conv_level(inplanes, planes, convs, stride=1):
        mods = []
        for i in range(convs):
            mods += [hnn.Conv2d(inplanes, planes, 3, stride if i == 0 else 1, 1), hnn.BatchNorm2d(planes), nn.Identity()]
            inplanes = planes
        return nn.Sequential(*mods)

    @staticmethod
    def _takes_raw(conv):
        """a conv whose kernels can apply the previous layer's BN + ReLU on load (16 input channels, 3x3 / pad 1)"""
        return isinstance(conv, hnn.Conv2d) and conv.k == 3 and conv.padding == 1 and conv.weight.shape[1] == 16 and conv.bias is None

    @classmethod
    def _run_conv_level(cls, seq, x, next_conv=None):
        """next_conv: the single consumer of this level's output when it is a conv that takes a raw input (the BN apply pass of the
        level's last layer is then left to it)"""
        for i in range(0, len(seq), 3):
            nxt = seq[i + 3] if i + 3 < len(seq) else next_conv
            x = hnn.conv_bn_act(seq[i], seq[i + 1], x, defer=nxt is not None and cls._takes_raw(nxt))
        return x

    def forward(self, img):
        # base_layer -> level0 -> level1 are plain conv -> BN -> ReLU chains on the two largest tensors of the network (16 channels
        # at full resolution): in training their normalised activations are never stored — the next conv applies the BN on load
        mixed = self.fp32_levels if self.compute_dtype != torch.float32 else 0
        x = hnn.stem_bn_act(self.base_layer[0], self.base_layer[1], img, torch.float32 if mixed else self.compute_dtype,
                            defer=self._takes_raw(self.level0[0]))
        y = []
        for i in range(6):
            level = getattr(self, f"level{i}")
            if mixed and i == mixed:
                x = x.to(self.compute_dtype)        # the fp32 prefix ends here
            x = self._run_conv_level(level, x, self.level1[0] if i == 0 and not self.expose_level0 else None) if i < 2 else level(x)
            if i >= 2:
                x = ops.share(x)        # a level's output feeds the next level and the up path
            # a deferred level0 output is a RAW conv output (its BN + ReLU live in level1's first conv): never hand that out
            y.append(None if getattr(x, "_cn_pre", None) is not None else x)
        if mixed:
            y = [t if t is None or t.dtype == self.compute_dtype else t.to(self.compute_dtype) for t in y]
        return y


def dla34(pretrained=False, **kwargs):
    """pose_dla_dcn.py:400-406 (no download: `pretrained` must be False or a local .pth path)."""
    model = DLA([1, 1, 1, 2, 2, 1], [16, 32, 64, 128, 256, 512], block=BasicBlock, **kwargs)
    if pretrained:
        if not isinstance(pretrained, str):
            raise RuntimeError("no network: pass a local path to the ImageNet dla34 weights instead of pretrained=True")
        sd = torch.load(pretrained, map_location="cpu")
        model.load_state_dict({k: v for k, v in sd.items() if not k.startswith("fc.")})
    return model


class DeformConv(nn.Module):
    """pose_dla_dcn.py:435-454: DCN -> BN -> ReLU."""

    def __init__(self, chi, cho):
        super().__init__()
        self.actf = nn.Sequential(hnn.BatchNorm2d(cho), nn.Identity())
        self.conv = hnn.DCN(chi, cho, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1)

    def forward(self, x):
        bn = self.actf[0]
        if not bn.training and not (torch.is_grad_enabled() and (self.conv.weight.requires_grad or x.requires_grad)):
            s, b = bn.folded()
            return self.conv.infer(x, s, b, True)
        return bn(self.conv(x, bn_stats=bn.training), None, True)


class IDAUp(nn.Module):
    """pose_dla_dcn.py:457-488."""

    def __init__(self, o, channels, up_f):
        super().__init__()
        for i in range(1, len(channels)):
            setattr(self, f"proj_{i}", DeformConv(channels[i], o))
            setattr(self, f"node_{i}", DeformConv(o, o))
            setattr(self, f"up_{i}", hnn.DepthwiseUp(o, int(up_f[i])))

    def forward(self, layers, startp, endp):
        for i in range(startp + 1, endp):
            j = i - startp
            merged = getattr(self, f"up_{j}")(getattr(self, f"proj_{j}")(layers[i]), layers[i - 1])    # up(proj(x)) + layers[i-1]
            layers[i] = ops.share(getattr(self, f"node_{j}")(merged))     # later stages read it as DCN input and as skip operand


class DLAUp(nn.Module):
    """pose_dla_dcn.py:491-516."""

    def __init__(self, startp, channels, scales, in_channels=None):
        super().__init__()
        self.startp = startp
        channels = list(channels)
        in_channels = list(channels) if in_channels is None else list(in_channels)
        scales = [int(s) for s in scales]
        for i in range(len(channels) - 1):
            j = -i - 2
            setattr(self, f"ida_{i}", IDAUp(channels[j], in_channels[j:], [s // scales[j] for s in scales[j:]]))
            scales[j + 1:] = [scales[j]] * len(scales[j + 1:])
            in_channels[j + 1:] = [channels[j]] * len(in_channels[j + 1:])

    def forward(self, layers):
        layers = list(layers)
        out = [layers[-1]]
        for i in range(len(layers) - self.startp - 1):
            getattr(self, f"ida_{i}")(layers, len(layers) - i - 2, len(layers))
            out.insert(0, layers[-1])
        return out


class DLASeg(nn.Module):
    """pose_dla_dcn.py:532-570.  Returns [NHWC feature map with `out_channels` channels at stride `down_ratio`]."""

    def __init__(self, base_name, pretrained, down_ratio, final_kernel, last_level, out_channel=0,
                 compute_dtype=torch.bfloat16):
        super().__init__()
        assert down_ratio in [2, 4, 8, 16]
        self.first_level = int(math.log2(down_ratio))
        self.last_level = last_level
        self.nchw_out = False                 # True: return the reference's NCHW fp32 map instead of the NHWC handle
        self.base = globals()[base_name](pretrained=pretrained, compute_dtype=compute_dtype)
        self.base.expose_level0 = self.first_level == 0      # y[0] is read only when the up path starts at stride 1
        channels = self.base.channels
        scales = [2 ** i for i in range(len(channels[self.first_level:]))]
        self.dla_up = DLAUp(self.first_level, channels[self.first_level:], scales)
        self.out_channels = out_channel if out_channel else channels[self.first_level]
        self.ida_up = IDAUp(self.out_channels, channels[self.first_level:self.last_level],
                            [2 ** i for i in range(self.last_level - self.first_level)])

    @property
    def compute_dtype(self):
        return self.base.compute_dtype

    @compute_dtype.setter
    def compute_dtype(self, dt):
        self.base.compute_dtype = dt

    def forward(self, img):
        x = self.dla_up(self.base(img))
        y = [x[i] for i in range(self.last_level - self.first_level)]   # .clone() of the reference: tensors are never mutated here
        self.ida_up(y, 0, len(y))
        return ops.emit_maps([y[-1]], self.out_channels, self.nchw_out)


def get_pose_net(num_layers, down_ratio=4, compute_dtype=torch.bfloat16, pretrained=False):
    """pose_dla_dcn.py:573-581 without the ImageNet download."""
    return DLASeg(f"dla{num_layers}", pretrained=pretrained, down_ratio=down_ratio, final_kernel=1, last_level=5,
                  compute_dtype=compute_dtype)
