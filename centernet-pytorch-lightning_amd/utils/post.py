"""Test-time augmentation and detection post-processing on the device (reference: CenterNetDetection.test_step /
test_step_end, centernet_detection.py:132-225; soft-NMS utils/nms.py:5-107).

The reference prepares one image at a time with torchvision, syncs with `.cpu()` after every decode and loops over the 80
classes in Python.  These wrappers keep a whole batch on the GPU: `tta_prepare` (pad + normalise + mirror in one launch),
`flip_merge` (average the mirrored head maps), `ctdet_merge` (map boxes back to the image, group by class, multi-scale
soft-NMS, max-per-image cut) — the host gets ONE tensor per batch and only formats it (`results_by_class`).
"""
import numpy as np
import torch

from .._hip import call


def tta_pad(size, padding):
    """centernet_detection.py:143-144: pad each side so that the padded size is `(size | padding) + 1`."""
    return ((size | padding) + 1 - size) // 2


def tta_prepare(img, mean, std, pad_x, pad_y, flip):
    """img fp32 [B,3,H,W] in [0,1] (already resized) -> fp32 [B*(1+flip),3,H+2pad_y,W+2pad_x]; rows B.. are mirrored."""
    B, _, H, W = img.shape
    out = torch.empty((B * (2 if flip else 1), 3, H + 2 * pad_y, W + 2 * pad_x), dtype=torch.float32, device=img.device)
    call("cn_tta_prepare", img.float().contiguous(), out, B, H, W, int(pad_x), int(pad_y), float(mean[0]), float(mean[1]),
         float(mean[2]), float(std[0]), float(std[1]), float(std[2]), int(bool(flip)))
    return out


def tta_prepare_scaled(img, new_h, new_w, mean, std, pad_x, pad_y, flip):
    """img fp32 [B,3,H,W] in [0,1] at its original size -> resized to (new_h, new_w) (VF.resize on a tensor: bilinear, no
    antialias), padded, normalised and mirrored in ONE launch -> fp32 [B*(1+flip),3,new_h+2pad_y,new_w+2pad_x]."""
    B, _, H, W = img.shape
    out = torch.empty((B * (2 if flip else 1), 3, new_h + 2 * pad_y, new_w + 2 * pad_x), dtype=torch.float32, device=img.device)
    call("cn_tta_prepare_scaled", img.float().contiguous(), out, B, H, W, int(new_h), int(new_w), int(pad_x), int(pad_y),
         float(mean[0]), float(mean[1]), float(mean[2]), float(std[0]), float(std[1]), float(std[2]), int(bool(flip)))
    return out


def flip_merge(x):
    """x fp32 [2B,C,H,W] -> (x[:B] + hflip(x[B:])) / 2."""
    B2, C, H, W = x.shape
    out = torch.empty((B2 // 2, C, H, W), dtype=torch.float32, device=x.device)
    call("cn_flip_merge", x.contiguous(), out, B2 // 2, C, H, W)
    return out


def flip_merge_perm(x, perm, sign):
    """x fp32 [2B,C,H,W] -> (x[:B] + sign[c] * hflip(x[B:, perm[c]])) / 2 (pose-aware merges, centernet_multi_pose.py:200-211)."""
    B2, C, H, W = x.shape
    out = torch.empty((B2 // 2, C, H, W), dtype=torch.float32, device=x.device)
    call("cn_flip_merge_perm", x.contiguous(), out, perm, sign, B2 // 2, C, H, W)
    return out


def pose_merge(dets, metas, down_ratio=4, max_per_image=20, nms_method=2, nms_nt=0.5, nms_sigma=0.5, nms_threshold=0.001):
    """dets: list (one per test scale) of multi_pose_decode outputs [B,K,57] -> rows fp32 [B, S*K, 57], counts int32 [B]."""
    S = len(dets)
    B, K, D = dets[0].shape
    d = torch.stack([t.float() for t in dets]).contiguous()
    meta = torch.tensor([[m["padding"][0], m["padding"][1], m["scale"][0], m["scale"][1]] for m in metas], dtype=torch.float32,
                        device=d.device)
    rows = torch.empty((B, S * K, D), dtype=torch.float32, device=d.device)
    counts = torch.empty((B,), dtype=torch.int32, device=d.device)
    call("cn_pose_merge", d, meta, rows, counts, S, B, K, D, int(down_ratio), int(max_per_image), int(nms_method), float(nms_nt),
         float(nms_sigma), float(nms_threshold))
    return rows, counts


def ctdet_merge(dets, metas, num_classes, down_ratio=4, max_per_image=100, nms_method=2, nms_nt=0.5, nms_sigma=0.5,
                nms_threshold=0.001):
    """dets: list (one per test scale) of ctdet_decode outputs [B,K,6]; metas: list of {"scale": [sx, sy], "padding": [px, py]}.
    -> rows fp32 [B, S*K, 6] (class-ascending, zero padded), counts int32 [B]."""
    S = len(dets)
    B, K, _ = dets[0].shape
    d = torch.stack([t.float() for t in dets]).contiguous()
    meta = torch.tensor([[m["padding"][0], m["padding"][1], m["scale"][0], m["scale"][1]] for m in metas], dtype=torch.float32,
                        device=d.device)
    rows = torch.empty((B, S * K, 6), dtype=torch.float32, device=d.device)
    counts = torch.empty((B,), dtype=torch.int32, device=d.device)
    call("cn_ctdet_merge", d, meta, rows, counts, S, B, K, int(num_classes), int(down_ratio), int(max_per_image), int(nms_method),
         float(nms_nt), float(nms_sigma), float(nms_threshold))
    return rows, counts


def results_by_class(rows, counts, num_classes):
    """One device->host copy per batch, then the reference's result format: per image {class_id (1-based): ndarray [n, 5]}."""
    rows, counts = rows.cpu().numpy(), counts.cpu().numpy()
    out = []
    for b in range(rows.shape[0]):
        r = rows[b, :counts[b]]
        cls = r[:, 5].astype(np.int64)
        out.append({j + 1: r[cls == j, :5].reshape(-1, 5) for j in range(num_classes)})
    return out
