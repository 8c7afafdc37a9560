"""ORACLE (test infrastructure): torch-CPU restatement of the reference backbones and heads.

Same op sequence and the same ``state_dict`` names as the reference, so weights move freely
between the imported reference (build container only), this oracle and the HIP modules.
DLA uses ``oracle.dcn_ref.DCN`` (DCNv2 parity is unpinned, see that file).
"""
import math

import torch
import torch.nn as nn

from .dcn_ref import DCN

MOM = 0.1  # BN_MOMENTUM: msra_resnet.py:11, pose_dla_dcn.py:13


def _bn(c):
    return nn.BatchNorm2d(c, momentum=MOM)


# ----------------------------------------------------------------------------- ResNet + deconv
class ResBasic(nn.Module):
    """msra_resnet.py:29-58."""
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = _bn(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = _bn(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idt)


class ResBottleneck(nn.Module):
    """msra_resnet.py:61-100."""
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = _bn(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = _bn(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = _bn(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


RESNET_SPEC = {18: (ResBasic, [2, 2, 2, 2]), 34: (ResBasic, [3, 4, 6, 3]), 50: (ResBottleneck, [3, 4, 6, 3]),
               101: (ResBottleneck, [3, 4, 23, 3]), 152: (ResBottleneck, [3, 8, 36, 3])}


class PoseResNet(nn.Module):
    """msra_resnet.py:103-207: stem 7x7/s2 + maxpool 3x3/s2 + 4 stages + 3x(ConvT 4x4/s2 -> BN -> ReLU)."""

    def __init__(self, num_layers=18):
        super().__init__()
        block, layers = RESNET_SPEC[num_layers]
        self.out_channels = 256
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = _bn(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._stage(block, 64, layers[0], 1)
        self.layer2 = self._stage(block, 128, layers[1], 2)
        self.layer3 = self._stage(block, 256, layers[2], 2)
        self.layer4 = self._stage(block, 512, layers[3], 2)
        up = []
        for _ in range(3):
            up += [nn.ConvTranspose2d(self.inplanes, 256, 4, 2, 1, 0, bias=False), _bn(256), nn.ReLU(inplace=True)]
            self.inplanes = 256
        self.deconv_layers = nn.Sequential(*up)

    def _stage(self, block, planes, n, stride):
        ds = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                               _bn(planes * block.expansion))
        mods = [block(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * block.expansion
        mods += [block(self.inplanes, planes) for _ in range(1, n)]
        return nn.Sequential(*mods)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return [self.deconv_layers(x)]


class PoseResNetDCN(PoseResNet):
    """resnet_dcn.py:131-261: the same ResNet trunk, up path 3x(DCN 3x3 -> BN -> ReLU -> ConvT 4x4/s2 -> BN -> ReLU) with
    256/128/64 filters (resnet_dcn.py:146-150, 189-234); out_channels = 64 (resnet_dcn.py:134)."""

    def __init__(self, num_layers=18):
        super().__init__(num_layers)
        from .dcn_ref import DCN
        block, _ = RESNET_SPEC[num_layers]
        self.out_channels = 64
        self.inplanes = 512 * block.expansion
        up = []
        for planes in (256, 128, 64):
            up += [DCN(self.inplanes, planes), _bn(planes), nn.ReLU(inplace=True),
                   nn.ConvTranspose2d(planes, planes, 4, 2, 1, 0, bias=False), _bn(planes), nn.ReLU(inplace=True)]
            self.inplanes = planes
        self.deconv_layers = nn.Sequential(*up)


# ----------------------------------------------------------------------------- DLA-34 + DCN up path
class DlaBasic(nn.Module):
    """pose_dla_dcn.py:28-68 (residual is added before the last ReLU)."""

    def __init__(self, cin, cout, stride=1, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, dilation, dilation, bias=False)
        self.bn1 = _bn(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, dilation, dilation, bias=False)
        self.bn2 = _bn(cout)

    def forward(self, x, residual=None):
        if residual is None:
            residual = x
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + residual)


class DlaRoot(nn.Module):
    """pose_dla_dcn.py:165-188."""

    def __init__(self, cin, cout, kernel_size, residual):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 1, 1, (kernel_size - 1) // 2, bias=False)
        self.bn = _bn(cout)
        self.relu = nn.ReLU(inplace=True)
        self.residual = residual

    def forward(self, *xs):
        y = self.bn(self.conv(torch.cat(xs, 1)))
        if self.residual:
            y = y + xs[0]
        return self.relu(y)


class DlaTree(nn.Module):
    """pose_dla_dcn.py:191-265 (incl. the computed-but-unused ``project`` of outer trees, :252-258)."""

    def __init__(self, levels, block, cin, cout, stride=1, level_root=False, root_dim=0,
                 root_kernel_size=1, dilation=1, root_residual=False):
        super().__init__()
        if root_dim == 0:
            root_dim = 2 * cout
        if level_root:
            root_dim += cin
        if levels == 1:
            self.tree1 = block(cin, cout, stride, dilation=dilation)
            self.tree2 = block(cout, cout, 1, dilation=dilation)
            self.root = DlaRoot(root_dim, cout, root_kernel_size, root_residual)
        else:
            self.tree1 = DlaTree(levels - 1, block, cin, cout, stride, root_dim=0,
                                 root_kernel_size=root_kernel_size, dilation=dilation, root_residual=root_residual)
            self.tree2 = DlaTree(levels - 1, block, cout, cout, root_dim=root_dim + cout,
                                 root_kernel_size=root_kernel_size, dilation=dilation, root_residual=root_residual)
        self.level_root, self.root_dim, self.levels = level_root, root_dim, levels
        self.downsample = nn.MaxPool2d(stride, stride=stride) if stride > 1 else None
        self.project = None
        if cin != cout:
            self.project = nn.Sequential(nn.Conv2d(cin, cout, 1, 1, bias=False), _bn(cout))

    def forward(self, x, residual=None, children=None):
        children = [] if children is None else children
        bottom = self.downsample(x) if self.downsample else x
        residual = self.project(bottom) if self.project else bottom
        if self.level_root:
            children.append(bottom)
        x1 = self.tree1(x, residual)
        if self.levels == 1:
            return self.root(self.tree2(x1), x1, *children)
        children.append(x1)
        return self.tree2(x1, children=children)


class DlaBase(nn.Module):
    """pose_dla_dcn.py:268-378."""

    def __init__(self, levels, channels, block=DlaBasic):
        super().__init__()
        self.channels = channels
        self.base_layer = nn.Sequential(nn.Conv2d(3, channels[0], 7, 1, 3, bias=False), _bn(channels[0]),
                                        nn.ReLU(inplace=True))
        self.level0 = self._convs(channels[0], channels[0], levels[0])
        self.level1 = self._convs(channels[0], channels[1], levels[1], stride=2)
        self.level2 = DlaTree(levels[2], block, channels[1], channels[2], 2, level_root=False)
        self.level3 = DlaTree(levels[3], block, channels[2], channels[3], 2, level_root=True)
        self.level4 = DlaTree(levels[4], block, channels[3], channels[4], 2, level_root=True)
        self.level5 = DlaTree(levels[5], block, channels[4], channels[5], 2, level_root=True)

    @staticmethod
    def _convs(cin, cout, n, stride=1):
        mods = []
        for i in range(n):
            mods += [nn.Conv2d(cin, cout, 3, stride if i == 0 else 1, 1, bias=False), _bn(cout), nn.ReLU(inplace=True)]
            cin = cout
        return nn.Sequential(*mods)

    def forward(self, x):
        x = self.base_layer(x)
        outs = []
        for i in range(6):
            x = getattr(self, f"level{i}")(x)
            outs.append(x)
        return outs


def bilinear_up_weights_(w):
    """pose_dla_dcn.py:424-432."""
    k = w.size(2)
    f = math.ceil(k / 2)
    c = (2 * f - 1 - f % 2) / (2.0 * f)
    g = torch.tensor([1 - abs(i / f - c) for i in range(k)], dtype=w.dtype)
    w.copy_(torch.outer(g, g).expand_as(w))


class DeformConv(nn.Module):
    """pose_dla_dcn.py:435-454: DCN -> BN -> ReLU."""

    def __init__(self, chi, cho):
        super().__init__()
        self.actf = nn.Sequential(_bn(cho), nn.ReLU(inplace=True))
        self.conv = DCN(chi, cho, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1)

    def forward(self, x):
        return self.actf(self.conv(x))


class IDAUp(nn.Module):
    """pose_dla_dcn.py:457-488."""

    def __init__(self, o, channels, up_f):
        super().__init__()
        for i in range(1, len(channels)):
            f = int(up_f[i])
            setattr(self, f"proj_{i}", DeformConv(channels[i], o))
            setattr(self, f"node_{i}", DeformConv(o, o))
            up = nn.ConvTranspose2d(o, o, f * 2, stride=f, padding=f // 2, output_padding=0, groups=o, bias=False)
            with torch.no_grad():
                bilinear_up_weights_(up.weight)
            setattr(self, f"up_{i}", up)

    def forward(self, layers, startp, endp):
        for i in range(startp + 1, endp):
            j = i - startp
            layers[i] = getattr(self, f"up_{j}")(getattr(self, f"proj_{j}")(layers[i]))
            layers[i] = getattr(self, f"node_{j}")(layers[i] + layers[i - 1])


class DLAUp(nn.Module):
    """pose_dla_dcn.py:491-516."""

    def __init__(self, startp, channels, scales):
        super().__init__()
        self.startp = startp
        channels = list(channels)
        in_channels = list(channels)
        scales = list(scales)
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
    """pose_dla_dcn.py:532-570 with dla34 = DLA([1,1,1,2,2,1],[16,32,64,128,256,512]) (:400-403),
    down_ratio=4, last_level=5 (:573-581)."""

    def __init__(self, down_ratio=4, last_level=5):
        super().__init__()
        self.first_level = int(math.log2(down_ratio))
        self.last_level = last_level
        self.base = DlaBase([1, 1, 1, 2, 2, 1], [16, 32, 64, 128, 256, 512])
        ch = self.base.channels
        scales = [2 ** i for i in range(len(ch[self.first_level:]))]
        self.dla_up = DLAUp(self.first_level, ch[self.first_level:], scales)
        self.out_channels = ch[self.first_level]
        self.ida_up = IDAUp(self.out_channels, ch[self.first_level:self.last_level],
                            [2 ** i for i in range(self.last_level - self.first_level)])

    def forward(self, x):
        x = self.dla_up(self.base(x))
        y = [x[i].clone() for i in range(self.last_level - self.first_level)]
        self.ida_up(y, 0, len(y))
        return [y[-1]]


# ----------------------------------------------------------------------------- Hourglass-104
class HgConvolution(nn.Module):
    """large_hourglass.py:11-31 (default BN momentum, unlike the ResNet/DLA files)."""

    def __init__(self, k, cin, cout, stride=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, bias=False)
        self.bn = nn.BatchNorm2d(cout)

    def forward(self, x):
        return torch.relu(self.bn(self.conv(x)))


class HgResidual(nn.Module):
    """large_hourglass.py:52-92."""

    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.skip = (nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))
                     if stride != 1 or cin != cout else nn.Sequential())

    def forward(self, x):
        y = torch.relu(self.bn1(self.conv1(x)))
        return torch.relu(self.bn2(self.conv2(y)) + self.skip(x))


def _hg_layer(cin, cout, n, stride=1):      # make_layer :95-99, make_hg_layer :325-328
    return nn.Sequential(HgResidual(cin, cout, stride), *[HgResidual(cout, cout) for _ in range(n - 1)])


def _hg_layer_revr(cin, cout, n):           # make_layer_revr :102-107
    return nn.Sequential(*[HgResidual(cin, cin) for _ in range(n - 1)], HgResidual(cin, cout))


class HgModule(nn.Module):
    """kp_module, large_hourglass.py:143-204: up1(x) + nearest_up2x(low3(low2(low1(x)))); pool layer is empty (:120-121)."""

    def __init__(self, n, dims, mods):
        super().__init__()
        self.up1 = _hg_layer(dims[0], dims[0], mods[0])
        self.low1 = _hg_layer(dims[0], dims[1], mods[0], stride=2)
        self.low2 = HgModule(n - 1, dims[1:], mods[1:]) if n > 1 else _hg_layer(dims[1], dims[1], mods[1])
        self.low3 = _hg_layer_revr(dims[1], dims[0], mods[0])

    def forward(self, x):
        low = self.low3(self.low2(self.low1(x)))
        return self.up1(x) + nn.functional.interpolate(low, scale_factor=2, mode="nearest")


class Hourglass(nn.Module):
    """exkp :207-322 with HourglassNet's constants :331-348; returns one 256-channel map per stack."""

    def __init__(self, nstack=2):
        super().__init__()
        dims, mods = [256, 256, 384, 384, 384, 512], [2, 2, 2, 2, 2, 4]
        self.nstack, self.out_channels = nstack, 256
        self.pre = nn.Sequential(HgConvolution(7, 3, 128, 2), HgResidual(128, 256, 2))
        self.kps = nn.ModuleList([HgModule(5, dims, mods) for _ in range(nstack)])
        self.cnvs = nn.ModuleList([HgConvolution(3, 256, 256) for _ in range(nstack)])
        self.inters = nn.ModuleList([HgResidual(256, 256) for _ in range(nstack - 1)])
        self.inters_ = nn.ModuleList([nn.Sequential(nn.Conv2d(256, 256, 1, bias=False), nn.BatchNorm2d(256)) for _ in range(nstack - 1)])
        self.cnvs_ = nn.ModuleList([nn.Sequential(nn.Conv2d(256, 256, 1, bias=False), nn.BatchNorm2d(256)) for _ in range(nstack - 1)])

    def forward(self, x):
        inter, outs = self.pre(x), []
        for i in range(self.nstack):
            cnv = self.cnvs[i](self.kps[i](inter))
            outs.append(cnv)
            if i < self.nstack - 1:
                inter = self.inters[i](torch.relu(self.inters_[i](inter) + self.cnvs_[i](cnv)))
        return outs




# ----------------------------------------------------------------------------- heads
class HeadConv(nn.Module):
    """heads.py:4-25."""

    def __init__(self, out_channels, intermediate_channel, head_conv):
        super().__init__()
        self.out_channels = out_channels
        self.fc = nn.Sequential(nn.Conv2d(intermediate_channel, head_conv, 3, padding=1, bias=True),
                                nn.ReLU(inplace=True), nn.Conv2d(head_conv, out_channels, 1))

    def forward(self, x):
        return self.fc(x)


class CenterHead(nn.Module):
    """heads.py:28-50."""

    def __init__(self, heads, intermediate_channel, head_conv):
        super().__init__()
        self.heads = heads
        for name, c in heads.items():
            setattr(self, name, HeadConv(c, intermediate_channel, head_conv))
            h = getattr(self, name)
            if name.startswith("heatmap"):
                h.fc[-1].bias.data.fill_(-2.19)
            else:
                for m in h.modules():
                    if isinstance(m, nn.Conv2d):
                        nn.init.normal_(m.weight, std=0.001)
                        nn.init.constant_(m.bias, 0)

    def forward(self, x):
        return {name: getattr(self, name)(x) for name in self.heads}


def create_model(arch):
    """models/__init__.py:14-19 without the weight download."""
    name, _, n = arch.partition("_")
    if name == "res":
        return PoseResNet(int(n))
    if name == "resdcn":
        return PoseResNetDCN(int(n))
    if name == "dla":
        assert int(n) == 34
        return DLASeg()
    if name == "hourglass":
        return Hourglass()
    raise KeyError(arch)


CTDET_HEADS = {"heatmap": 80, "width_height": 2, "regression": 2}                       # centernet_detection.py:59
POSE_HEADS = {"heatmap": 1, "width_height": 2, "regression": 2, "heatmap_keypoints": 17,
              "keypoints": 34, "heatmap_keypoints_offset": 2}                            # centernet_multi_pose.py:54-61


class CenterNetRef(nn.Module):
    """backbone + heads + loss, as centernet_detection.py:43-130 / centernet_multi_pose.py:36-155."""

    def __init__(self, arch, heads=None, task="ctdet"):
        super().__init__()
        from . import ops_ref
        self.backbone = create_model(arch)
        head_conv = 256 if "dla" in arch or "hourglass" in arch else 64                  # centernet.py:15
        num_stacks = 2 if "hourglass" in arch else 1                                     # centernet.py:16
        self.task = task
        heads = dict(heads or (CTDET_HEADS if task == "ctdet" else POSE_HEADS))
        self.heads = nn.ModuleList([CenterHead(heads, self.backbone.out_channels, head_conv) for _ in range(num_stacks)])
        self._ops = ops_ref

    def forward(self, x):
        return [h(o) for h, o in zip(self.heads, self.backbone(x))]

    def loss(self, outputs, target):
        fn = self._ops.ctdet_loss if self.task == "ctdet" else self._ops.multi_pose_loss
        if len(outputs) == 1:
            return fn(outputs[0], target)
        # centernet_detection.py:99-123: per-stack terms are SUMMED, the weighted total is divided by num_stacks
        per = [fn(o, target)[1] for o in outputs]
        stats = {k: sum(p[k] for p in per) for k in per[0]}
        stats["loss"] = stats["loss"] / len(outputs)
        return stats["loss"], stats
