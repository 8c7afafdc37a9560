"""Loss modules with the reference's names (CenterNet/utils/losses.py) on the HIP loss kernels (csrc/loss.hip)."""
import torch
import torch.nn as nn

from .. import ops


def _neg_loss(pred, gt):
    """utils/losses.py:14-39 (penalty-reduced focal loss); `num_pos == 0` is resolved on the device (no host sync)."""
    return ops.FocalLossFn.apply(pred, gt)


class FocalLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.neg_loss = _neg_loss

    def forward(self, out, target):
        return self.neg_loss(out, target)

    def on_logits(self, x, target, clamp=1e-4):
        """`y = sigmoid_clamped(x); loss = self(y, target)` (centernet_detection.py:103-106) as one autograd node with a
        single-pass backward.  Returns (y, loss); x holds sigmoid(x) afterwards, exactly as after sigmoid_clamped."""
        if self.neg_loss is not _neg_loss or not x.is_cuda or x.dtype != torch.float32 or not x.is_contiguous():
            from .decode import sigmoid_clamped
            y = sigmoid_clamped(x, clamp)
            return y, self(y, target)
        _, y, loss = ops.SigmoidFocalFn.apply(x, target, clamp)
        return y, loss


class RegL1Loss(nn.Module):
    """utils/losses.py:53-63: mask [B,N] is broadcast over the channels."""

    def forward(self, output, mask, ind, target):
        return ops.GatherL1Fn.apply(output, mask, ind, target)


class RegWeightedL1Loss(nn.Module):
    """utils/losses.py:81-91: mask [B,N,C]."""

    def forward(self, output, mask, ind, target):
        return ops.GatherL1Fn.apply(output, mask, ind, target)


class _NormRegL1Fn(torch.autograd.Function):
    """Rows gathered by cn_gather_rows; the arithmetic on the [B,N,C] rows (a few thousand elements) and the scatter of their
    gradient are plain tensor ops — this loss is not on any path the reference's task modules take."""

    @staticmethod
    def forward(ctx, output, mask, ind, target):
        from .._hip import call
        feat = output.contiguous().float()
        B, C = feat.shape[:2]
        HW = feat[0, 0].numel()
        ind = ind.contiguous().long()
        N = ind.shape[1]
        pred = torch.empty((B, N, C), dtype=torch.float32, device=feat.device)
        call("cn_gather_rows", feat, ind, pred, B, C, HW, N)
        m = mask.unsqueeze(2).expand_as(pred).float()
        r = pred / (target + 1e-4)
        den = m.sum() + 1e-4
        ctx.save_for_backward(r, m, target, ind, den)
        ctx.shape, ctx.hw = feat.shape, HW
        return (r * m - m).abs().sum() / den

    @staticmethod
    def backward(ctx, g):
        r, m, target, ind, den = ctx.saved_tensors
        B, C = ctx.shape[:2]
        dpred = g * torch.sign(r * m - m) * m / (target + 1e-4) / den                 # [B,N,C]
        dfeat = torch.zeros((B, C, ctx.hw), dtype=torch.float32, device=r.device)
        dfeat.scatter_add_(2, ind.unsqueeze(1).expand(B, C, ind.shape[1]), dpred.permute(0, 2, 1).contiguous())
        return dfeat.view(ctx.shape), None, None, None


class NormRegL1Loss(nn.Module):
    """utils/losses.py:66-78: L1 between pred / (target + 1e-4) and 1 at the masked object slots (never instantiated by the
    reference's task modules)."""

    def forward(self, output, mask, ind, target):
        return _NormRegL1Fn.apply(output, mask, ind, target)
