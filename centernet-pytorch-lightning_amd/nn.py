"""Layer modules over the HIP operators.  Parameters keep the reference's names/shapes (NCHW fp32),
so `state_dict`s are interchangeable with the reference; activations flowing between layers are NHWC in the
backbone's compute dtype.

Training: conv -> (batch-stat BN + residual + ReLU) as separate HBM passes (statistics need the whole
conv output first).  Inference: BN is folded into the packed conv weights and residual/ReLU ride in the conv
epilogue, so a conv+BN+ReLU(+add) block is ONE kernel; folded weights are cached per parameter version.
"""
import math

import torch
import torch.nn as nn

import os

from . import ops

STEM_BN_FUSED = not (os.environ.get("CN_DISABLE_STEM_BN_FUSED") or os.environ.get("CN_DISABLE_WGRAD_C16"))   # stem conv + BN as one autograd node
# training-mode BN of the 16-channel 512^2 layers applied by the consuming conv: needs the kernels with the pre-affine hook, so the
# switches that turn those kernels off turn this off too (round-4 ADVICE: they used to end in CN_EUNSUPPORTED instead of the unfused chain)
BN_DEFER = not (os.environ.get("CN_DISABLE_BN_DEFER") or os.environ.get("CN_DISABLE_CONV_C16R") or os.environ.get("CN_DISABLE_WGRAD_C16"))


class Conv2d(nn.Module):
    def __init__(self, cin, cout, k, stride=1, padding=0, bias=False):
        super().__init__()
        self.stride, self.padding, self.k = stride, padding, k
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(cin * k * k)
            self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)
        self._cache = {}

    def extra_repr(self):
        co, ci, k, _ = self.weight.shape
        return f"{ci}, {co}, kernel_size={k}, stride={self.stride}, padding={self.padding}, bias={self.bias is not None}"

    def forward(self, x, relu=False, mask_dx=False, defer_relu_bwd=False, bn_stats=False):
        """bn_stats: the output goes straight into a training-mode BatchNorm2d (ops.BnStats: statistics from the conv epilogue)"""
        pre = getattr(x, "_cn_pre", None)           # x is a raw conv output whose BN (+ ReLU) is applied by this conv's kernels (ops.BnDeferFn)
        if torch.is_grad_enabled() and (self.weight.requires_grad or x.requires_grad):
            return ops.conv2d(x, self.weight, self.bias, self.stride, self.padding, relu, mask_dx, defer_relu_bwd, bn_stats=bn_stats, pre=pre)
        if pre is not None:
            c = self.infer_key(x)
            Co, _, KH, KW = self.weight.shape
            OH, OW = ops.conv_out(x.shape[1], KH, self.stride, self.padding), ops.conv_out(x.shape[2], KW, self.stride, self.padding)
            y = ops._igemm(x, c["wp"], c["b"], None, Co, KH, KW, self.stride, self.padding, False, relu, OH, OW, bn_stats=bn_stats, pre=pre)
            return ops.BnStats.pop(y) if bn_stats else y
        return self.infer(x, None, None, None, relu)

    def with_skip(self, x, bn_stats=False):
        """-> (conv(x), x) for a residual connection around this conv: the skip path's gradient is added in the epilogue of this
        conv's data-gradient kernel (training; otherwise just the pair)."""
        if torch.is_grad_enabled() and x.requires_grad and self.stride == 1:
            return ops.conv2d(x, self.weight, self.bias, self.stride, self.padding, False, False, False, True, bn_stats=bn_stats)
        return self.forward(x), x

    def infer_key(self, x, scale=None, shift=None):
        """-> the cache entry {wp: packed (folded) weights, b: folded bias} for x's dtype and the current parameter versions"""
        key = (x.dtype, self.weight._version, None if self.bias is None else self.bias._version, ops.WeightsEpoch.value,
               None if scale is None else (scale.data_ptr(), scale._version))
        hit = self._cache.get("k")
        if hit != key:
            wp = ops.pack_weight(self.weight, 1, x.dtype, scale)
            b = self.bias.detach() if self.bias is not None else None
            if shift is not None:
                b = shift if b is None else (b * scale + shift)   # tiny [Co] host-side fold (once per weight version)
            self._cache = {"k": key, "wp": wp, "b": b}
        return self._cache

    def infer(self, x, scale, shift, residual, relu, out_dtype=None):
        """No-grad path: y = act(conv(x)*scale + shift + residual); packed (folded) weights are cached."""
        c = self.infer_key(x, scale, shift)
        Co, _, KH, KW = self.weight.shape
        N, H, W, _ = x.shape
        OH, OW = ops.conv_out(H, KH, self.stride, self.padding), ops.conv_out(W, KW, self.stride, self.padding)
        return ops._igemm(x, c["wp"], c["b"], residual, Co, KH, KW, self.stride, self.padding, False, relu, OH, OW, out_dtype)

    def infer_nchw(self, x):
        """No-grad 1x1 conv + bias straight into the public fp32 NCHW layout (a head's last layer); packed weights cached as in `infer`."""
        self.infer_key(x)
        return ops.conv1x1_to_nchw(x, self._cache["wp"], self._cache["b"], self.weight.shape[0])


class BatchNorm2d(nn.BatchNorm2d):
    """Parameter/buffer holder with nn.BatchNorm2d's state_dict; the arithmetic lives in ops.BatchNormActFn."""

    def __init__(self, c):
        super().__init__(c, momentum=ops.BN_MOMENTUM)
        self._pending = 0
        self._fold = None

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        if self._pending:
            self.num_batches_tracked += self._pending
            self._pending = 0
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def folded(self):
        """(scale, shift) fp32 of the eval-mode affine; cached per buffer/parameter version and per `ops.WeightsEpoch` (the
        optimizer and the training-mode kernel update parameters / running statistics behind `_version`'s back)."""
        key = (self.weight._version, self.bias._version, self.running_mean._version, self.running_var._version,
               ops.WeightsEpoch.value)
        if self._fold is None or self._fold[0] != key:
            with torch.no_grad():
                scale = self.weight * torch.rsqrt(self.running_var + self.eps)   # [C] vectors, once per weight version
                shift = self.bias - self.running_mean * scale
            self._fold = (key, scale.contiguous(), shift.contiguous())
        return self._fold[1], self._fold[2]

    def forward(self, x, residual=None, relu=True, defer=False):
        """defer: the caller guarantees that the ONLY consumer of the result is a 3x3 Conv2d with 16 input channels (DLA base_layer ->
        level0 -> level1): the apply pass is then left to that conv's kernels (ops.BnDeferFn) when x carries its producer's statistics"""
        if self.training:
            self._pending += 1
            ops.WeightsEpoch.bump()              # running_mean / running_var are about to change through raw pointers
            if (defer and BN_DEFER and residual is None and x.dtype == torch.bfloat16 and x.shape[-1] == 16
                    and getattr(x, "_bn_part", None) is not None and getattr(x, "_cn_pre", None) is None):
                return ops.batch_norm_defer(x, self, relu)
            return ops.batch_norm_act(x, self, residual, relu)
        s, b = self.folded()
        return ops.scale_shift_act(x, s, b, residual, relu)


def conv_bn_act_skip(conv, bn, x):
    """(ReLU(BN(conv(x))), x) — the first half of a residual block whose identity path is its own input."""
    if bn.training and torch.is_grad_enabled() and x.requires_grad and conv.stride == 1:
        c, skip = conv.with_skip(x, bn_stats=True)
        return bn(c, None, True), skip
    return conv_bn_act(conv, bn, x), x


def cat_conv_bn_act(conv, bn, xs, residual=None, relu=True):
    """conv1x1(cat(xs, channel)) -> BN -> (+residual) -> ReLU without building the concatenation (DLA Root).  Falls back to
    concat + conv for anything that is not a bias-free 1x1 / stride-1 conv over 32-channel-aligned sources."""
    ok = (conv.k == 1 and conv.stride == 1 and conv.padding == 0 and conv.bias is None and 1 <= len(xs) <= 6
          and all(t.shape[-1] % 32 == 0 for t in xs))      # the data-gradient GEMM of a source reads rup32(C_s) operand rows
    if not ok or len(xs) == 1:
        return conv_bn_act(conv, bn, ops.concat(list(xs)), residual, relu)
    grad = torch.is_grad_enabled() and (conv.weight.requires_grad or any(t.requires_grad for t in xs))
    if bn.training or grad:
        return bn(ops.conv1x1_cat(xs, conv.weight, bn_stats=bn.training), residual, relu)
    s, b = bn.folded()                              # eval: BN folded into the packed weights, one launch
    key = ("cat", xs[0].dtype, conv.weight._version, ops.WeightsEpoch.value, s.data_ptr(), s._version)
    if conv._cache.get("k") != key:
        conv._cache = {"k": key, "wp": ops.pack_weight(conv.weight, 1, xs[0].dtype, s), "b": b}
    return ops.conv1x1_cat_raw([t.contiguous() for t in xs], conv._cache["wp"], conv._cache["b"], residual,
                               conv.weight.shape[0], relu)


def conv_bn_act(conv, bn, x, residual=None, relu=True, defer=False):
    """conv -> BN -> (+residual) -> ReLU.  One fused kernel in eval/no-grad mode.  defer: see BatchNorm2d.forward."""
    if bn.training:
        return bn(conv(x, bn_stats=True), residual, relu, defer)      # statistics from the conv kernel's epilogue where it has the hook
    if torch.is_grad_enabled() and (conv.weight.requires_grad or x.requires_grad):
        return bn(conv(x), residual, relu)          # eval-mode BN but gradients wanted: unfused affine pass
    s, b = bn.folded()
    return conv.infer(x, s, b, residual, relu)


class StemConv(nn.Module):
    """7x7 conv on the NCHW fp32 image -> NHWC activations in `dtype`."""

    def __init__(self, cin, cout, k, stride, padding):
        super().__init__()
        self.stride, self.padding = stride, padding
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))

    def forward(self, img, dtype, bn_stats=False):
        y = ops.StemConvFn.apply(img, self.weight, self.stride, self.padding, dtype, bn_stats)
        return ops.BnStats.pop(y) if bn_stats else y

    def infer(self, img, dtype, scale, shift, relu):
        """no-grad path with the following eval-mode BN (+ReLU) folded into the kernel's epilogue"""
        return ops.stem_conv_infer(img, self.weight, scale, shift, self.stride, self.padding, relu, dtype)


def stem_bn_act(stem, bn, img, dtype, relu=True, defer=False):
    """stem conv -> BN -> ReLU; one fused kernel in eval / no-grad mode.  defer: see BatchNorm2d.forward."""
    if (defer and BN_DEFER and STEM_BN_FUSED and bn.training and dtype == torch.bfloat16 and stem.weight.shape[0] == 16
            and stem.weight.shape[2] == 7 and stem.padding == 3 and stem.weight.shape[1] <= 3 and ops.BnStats.enabled):
        # the BN behind the stem lives entirely in the neighbouring kernels (ops.StemBnDeferFn)
        bn._pending += 1
        ops.WeightsEpoch.bump()
        return ops.stem_bn_defer(img, stem.weight, bn, stem.stride, stem.padding, dtype, relu)
    if bn.training or (torch.is_grad_enabled() and stem.weight.requires_grad):
        return bn(stem(img, dtype, bn_stats=bn.training), None, relu, defer)
    s, b = bn.folded()
    return stem.infer(img, dtype, s, b, relu)


class ConvTranspose2d(nn.Module):
    def __init__(self, cin, cout, k, stride, padding):
        super().__init__()
        self.stride, self.padding = stride, padding
        self.weight = nn.Parameter(torch.empty(cin, cout, k, k))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))

    def forward(self, x):
        return ops.conv_transpose2d(x, self.weight, self.stride, self.padding)


class DepthwiseUp(nn.Module):
    """ConvTranspose2d(o, o, 2f, stride=f, padding=f//2, groups=o, bias=False), bilinear init (pose_dla_dcn.py:424-432,466-476)."""

    def __init__(self, c, f):
        super().__init__()
        self.stride, self.padding = f, f // 2
        k = 2 * f
        fc = math.ceil(k / 2)
        cc = (2 * fc - 1 - fc % 2) / (2.0 * fc)
        g = torch.tensor([1 - abs(i / fc - cc) for i in range(k)])
        self.weight = nn.Parameter(torch.outer(g, g).expand(c, 1, k, k).clone())

    def forward(self, x, residual=None):
        """up(x) (+ residual, fused into the store)"""
        return ops.DwDeconvFn.apply(x, self.weight, self.stride, self.padding, residual)


class DCN(nn.Module):
    """Drop-in for DCN.dcn_v2.DCN(chi, cho, (3,3), stride=1, padding=1, dilation=1, deformable_groups=1)."""

    def __init__(self, cin, cout, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1):
        super().__init__()
        if tuple(kernel_size) != (3, 3) or stride != 1 or padding != 1 or dilation != 1 or deformable_groups != 1:
            raise NotImplementedError("DCN: only the 3x3/s1/p1/d1/g1 configuration the reference uses is implemented")
        self.weight = nn.Parameter(torch.empty(cout, cin, 3, 3))
        bound = 1.0 / math.sqrt(cin * 9)
        nn.init.uniform_(self.weight, -bound, bound)
        self.bias = nn.Parameter(torch.zeros(cout))
        self.conv_offset_mask = Conv2d(cin, 27, 3, 1, 1, bias=True)
        nn.init.zeros_(self.conv_offset_mask.weight)
        nn.init.zeros_(self.conv_offset_mask.bias)

        self._cache = {}

    def forward(self, x, bn_stats=False):
        y = ops.DCNv2Fn.apply(x, self.weight, self.bias, self.conv_offset_mask.weight, self.conv_offset_mask.bias, bn_stats)
        return ops.BnStats.pop(y) if bn_stats else y

    def infer(self, x, scale, shift, relu):
        """No-grad path with the following BN folded in: offset conv -> sampling -> ONE 1x1 GEMM (+shift, ReLU)."""
        key = (x.dtype, self.weight._version, self.bias._version, ops.WeightsEpoch.value, scale.data_ptr(), scale._version)
        if self._cache.get("k") != key:
            wp = ops.pack_weight(self.weight, 1, x.dtype, scale)
            self._cache = {"k": key, "wp": wp, "b": (self.bias.detach() * scale + shift).contiguous()}
        N, H, W, Ci = x.shape
        om = self.conv_offset_mask.infer(x, None, None, None, False, out_dtype=torch.float32)
        Co = self.weight.shape[0]
        cp = ops.rup(Co, 16)
        y = torch.empty((N, H, W, cp), dtype=x.dtype, device=x.device) if cp == Co else ops.zeros((N, H, W, cp), x.dtype, x.device)
        ops.call("cn_dcn_fwd", x, om, self._cache["wp"], self._cache["b"], y, N, H, W, Ci, Ci, Co, cp, om.shape[-1], int(relu),
                 ops.dtype_code(x.dtype))
        return y


class MaxPool2d(nn.Module):
    def __init__(self, k, stride, padding=0):
        super().__init__()
        self.k, self.stride, self.padding = k, stride, padding

    def forward(self, x):
        return ops.max_pool(x, self.k, self.stride, self.padding)
