"""multi_pose_decode (reference: CenterNet/decode/multi_pose.py:7-96): centre top-K, per-joint heat-map top-K and
the keypoint <-> heat-map-peak matching in one workgroup per image."""
import torch

from .. import _hip


def multi_pose_decode(heat, wh, kps, reg=None, hm_hp=None, hp_offset=None, K=100):
    """-> [B,K,4+1+2J+1+J] (bbox, score, keypoints, class, keypoint scores) — 57 columns for 17 COCO joints."""
    if hm_hp is None:
        raise ValueError("multi_pose_decode needs hm_hp (the reference raises NameError without it: multi_pose.py:94)")
    f = lambda t: t.contiguous().float() if t is not None else None
    heat, wh, kps, reg, hm_hp, hp_offset = map(f, (heat, wh, kps, reg, hm_hp, hp_offset))
    B, C, H, W = heat.shape
    if C != 1:
        raise NotImplementedError("multi_pose_decode: the centre heat map has one class (person)")
    J = kps.shape[1] // 2
    det = torch.empty((B, K, 5 + 2 * J + 1 + J), dtype=torch.float32, device=heat.device)
    n = _hip.query("cn_multi_pose_decode_workspace_bytes", B, J, K)
    ws = _hip.workspace(n, heat.device, "decode")
    _hip.call("cn_multi_pose_decode", heat, wh, kps, reg, hm_hp, hp_offset, det, B, J, H, W, K, ws, n)
    return det
