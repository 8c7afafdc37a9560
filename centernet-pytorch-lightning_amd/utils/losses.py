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


class NormRegL1Loss(nn.Module):
    """utils/losses.py:66-78 — never instantiated by the reference's task modules; not part of the hot path."""

    def forward(self, output, mask, ind, target):
        raise NotImplementedError("NormRegL1Loss is unused by the reference (SURVEY.md §2.1 #7) and not implemented")
