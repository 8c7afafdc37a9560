"""ctdet_decode (reference: CenterNet/decode/ctdet.py:6-38) as two fused HIP launches:
per-(image, class) 3x3-NMS + exact top-K, then per-image top-K over the C*K survivors + wh/reg gather + boxes."""
import torch

from .. import _hip


def ctdet_decode(heat, wh, reg=None, K=100, return_aux=False, logits_clamp=None):
    """heat [B,C,H,W] post-sigmoid, wh [B,2,H,W], reg [B,2,H,W] | None -> [B,K,6] = x1,y1,x2,y2,score,class.
    `return_aux` additionally returns the flat indices [B,K] int64 and classes [B,K] int32.
    logits_clamp = lo: `heat` holds LOGITS; the result is that of ctdet_decode(sigmoid_clamped(heat, lo), ...) — bit for bit — but the
    sigmoid is applied by the top-K kernel on load (cn_ctdet_decode_logits) and `heat` is left untouched."""
    heat, wh = heat.contiguous().float(), wh.contiguous().float()
    reg = reg.contiguous().float() if reg is not None else None
    B, C, H, W = heat.shape
    det = torch.empty((B, K, 6), dtype=torch.float32, device=heat.device)
    inds = torch.empty((B, K), dtype=torch.int64, device=heat.device) if return_aux else None
    clses = torch.empty((B, K), dtype=torch.int32, device=heat.device) if return_aux else None
    n = _hip.query("cn_ctdet_decode_workspace_bytes", B, C, K)
    ws = _hip.workspace(n, heat.device, "decode")
    if logits_clamp is not None:
        if not _hip.try_call("cn_ctdet_decode_logits", heat, wh, reg, det, inds, clses, B, C, H, W, K, float(logits_clamp), ws, n):
            from ..utils.decode import sigmoid_clamped          # shapes the streaming top-K does not take: the two-step form
            _hip.call("cn_ctdet_decode", sigmoid_clamped(heat.clone(), logits_clamp), wh, reg, det, inds, clses, B, C, H, W, K, ws, n)
        return (det, inds, clses) if return_aux else det
    _hip.call("cn_ctdet_decode", heat, wh, reg, det, inds, clses, B, C, H, W, K, ws, n)
    return (det, inds, clses) if return_aux else det
