"""Decode primitives with the reference's names (CenterNet/utils/decode.py), running on the HIP kernels of
csrc/decode.hip.  Inputs/outputs are the public NCHW fp32 tensors.  Tie rule of every top-k: score descending,
equal scores -> lower flat index first (torch.topk leaves it unspecified).  The hot path (ctdet_decode /
multi_pose_decode) does NOT go through these helpers — it uses the fused kernels."""
import torch

from .. import ops
from .._hip import call


def _f32c(t):
    return t.contiguous().float()


def _nms(heat, kernel=3):
    """utils/decode.py:5-10."""
    if kernel % 2 != 1:
        raise ValueError("an even window changes the pooled map's size: the reference's `hmax == heat` fails on it too")
    heat = _f32c(heat)
    B, C, H, W = heat.shape
    out = torch.empty_like(heat)
    if kernel == 3:
        call("cn_nms3x3", heat, out, B, C, H, W)
    else:
        call("cn_nms", heat, out, B, C, H, W, int(kernel))
    return out


def _topk_channel(scores, K=40):
    """utils/decode.py:31-40 -> (scores, inds, ys, xs), each [B,C,K]."""
    scores = _f32c(scores)
    B, C, H, W = scores.shape
    s = torch.empty((B, C, K), dtype=torch.float32, device=scores.device)
    i = torch.empty((B, C, K), dtype=torch.int32, device=scores.device)
    call("cn_topk_channel", scores, s, i, B, C, H, W, K, 0)
    i = i.long()
    return s, i, torch.div(i, W, rounding_mode="floor").float(), (i % W).float()


def _topk(scores, K=40):
    """utils/decode.py:13-28 -> (score, inds, clses, ys, xs), each [B,K]."""
    B, C, H, W = scores.shape
    s1, i1, ys1, xs1 = _topk_channel(scores, K)
    s2 = torch.empty((B, K), dtype=torch.float32, device=scores.device)
    j = torch.empty((B, K), dtype=torch.int32, device=scores.device)
    call("cn_topk_rows", s1.view(B, C * K), s2, j, B, C * K, K)
    j = j.long()
    pick = lambda t: torch.gather(t.view(B, C * K), 1, j)
    return s2, pick(i1), torch.div(j, K, rounding_mode="floor").int(), pick(ys1), pick(xs1)


def sigmoid_clamped(x, clamp=1e-4):
    """utils/decode.py:43-45: sigmoid in place on `x`, returns the clamped copy."""
    if not x.is_contiguous() or x.dtype != torch.float32:
        raise RuntimeError("sigmoid_clamped expects a contiguous fp32 head map")
    _, y = ops.SigmoidClampFn.apply(x, clamp)
    return y


def _gather_feat(feat, ind, mask=None):
    """utils/decode.py:48-56: feat [B,HW,C], ind [B,N] -> [B,N,C]."""
    out = torch.gather(feat, 1, ind.unsqueeze(2).expand(ind.size(0), ind.size(1), feat.size(2)))
    if mask is not None:
        out = out[mask.unsqueeze(2).expand_as(out)].view(-1, feat.size(2))
    return out


def _transpose_and_gather_feat(feat, ind):
    """utils/decode.py:59-63 without materialising the NHWC transpose: out[b,n,c] = feat[b,c,ind[b,n]]."""
    feat = _f32c(feat)
    B, C = feat.shape[:2]
    HW = feat[0, 0].numel()
    ind = ind.contiguous().long()
    out = torch.empty((B, ind.shape[1], C), dtype=torch.float32, device=feat.device)
    call("cn_gather_rows", feat, ind, out, B, C, HW, ind.shape[1])
    return out
