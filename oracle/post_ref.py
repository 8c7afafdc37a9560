"""ORACLE (test infrastructure): CPU restatement of the reference's test-time augmentation and detection post-processing.

Follows CenterNetDetection.test_step / test_step_end (centernet_detection.py:132-225) and utils/nms.py:5-107 (soft_nms).
PINNED: soft_nms / soft_nms_39 (tests/golden/soft_nms.npz: utils/nms.py run from the reference's own source; numba is absent, so
`jit` is shimmed to the identity and the boxes get float64 storage to keep numba's typing) and test_step_end /
pose_test_step_end (tests/golden/test_step_end.npz: the reference's method bodies compiled from their source files and run on
seeded two-scale head maps) — see oracle/gen_golden.py:gen_soft_nms / gen_test_step_end; tests/test_oracle_golden.py and
tests/test_gpu_post.py hold this file and the HIP path to both.
PARITY UNPINNED: resize, tta_prepare and flip_merge* (the test_step halves, centernet_detection.py:132-173) — they need
torchvision.transforms.functional (resize / normalize / hflip), which is absent; checked by known-answer tests only (`resize` is
stated as ATen's bilinear interpolate, which is what torchvision's tensor resize of the reference's era calls).
"""
import numpy as np
import torch
import torch.nn.functional as F


def tta_pad(size, padding):
    """centernet_detection.py:143-144."""
    return ((size | padding) + 1 - size) // 2


def resize(img, new_h, new_w):
    """centernet_detection.py:139-141: `VF.resize(img, (new_h, new_w))` on a tensor = bilinear interpolate, align_corners=False,
    no antialias (torchvision.transforms.functional_tensor.resize of the reference's era).  img fp32 [B,3,H,W]."""
    if (new_h, new_w) == tuple(img.shape[-2:]):
        return img
    return F.interpolate(img, size=(new_h, new_w), mode="bilinear", align_corners=False)


def tta_prepare(img, mean, std, pad_x, pad_y, flip):
    """centernet_detection.py:146-154: F.pad (zeros) -> VF.normalize -> cat with VF.hflip.  img fp32 [B,3,H,W]."""
    x = F.pad(img, (pad_x, pad_x, pad_y, pad_y))
    m = torch.tensor(mean, dtype=torch.float32).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, 3, 1, 1)
    x = (x - m) / s
    return torch.cat([x, x.flip(-1)]) if flip else x


def flip_merge(x):
    """centernet_detection.py:167-171 for B images: rows [B:] are the mirrored passes."""
    B = x.shape[0] // 2
    return (x[:B] + x[B:].flip(-1)) / 2


def soft_nms(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    """utils/nms.py:5-107, statement by statement.  `boxes` float32 [N,5], modified in place; returns the kept prefix length.
    numba types `float32 + 1` as float64, so the overlap arithmetic is carried in Python floats (double) here."""
    N = boxes.shape[0]
    for i in range(N):
        maxscore, maxpos = boxes[i, 4], i
        t = boxes[i].copy()
        pos = i + 1
        while pos < N:
            if maxscore < boxes[pos, 4]:
                maxscore, maxpos = boxes[pos, 4], pos
            pos += 1
        boxes[i] = boxes[maxpos]
        boxes[maxpos] = t
        tx1, ty1, tx2, ty2 = (float(v) for v in boxes[i, :4])
        pos = i + 1
        while pos < N:
            x1, y1, x2, y2 = (float(v) for v in boxes[pos, :4])
            area = (x2 - x1 + 1) * (y2 - y1 + 1)
            iw = min(tx2, x2) - max(tx1, x1) + 1
            if iw > 0:
                ih = min(ty2, y2) - max(ty1, y1) + 1
                if ih > 0:
                    ua = float((tx2 - tx1 + 1) * (ty2 - ty1 + 1) + area - iw * ih)
                    ov = iw * ih / ua
                    if method == 1:
                        weight = 1 - ov if ov > Nt else 1
                    elif method == 2:
                        weight = np.exp(-(ov * ov) / sigma)
                    else:
                        weight = 0 if ov > Nt else 1
                    boxes[pos, 4] = weight * float(boxes[pos, 4])
                    if boxes[pos, 4] < threshold:
                        boxes[pos] = boxes[N - 1]
                        N -= 1
                        pos -= 1
            pos += 1
    return N


def test_step_end(dets, metas, num_classes, down_ratio=4, max_per_image=100, multi_scale=None):
    """centernet_detection.py:173-225 for ONE image.  dets: list over scales of float32 [K,6] decode outputs; metas as the
    reference builds them.  Returns {class_id (1-based): ndarray [n,5]}."""
    multi_scale = len(dets) > 1 if multi_scale is None else multi_scale
    per_scale = []
    for det, meta in zip(dets, metas):
        det = torch.as_tensor(det, dtype=torch.float32).clone()
        padding = torch.tensor(meta["padding"] + meta["padding"], dtype=torch.float32)
        scale = torch.tensor(meta["scale"] + meta["scale"], dtype=torch.float32)
        det[:, :4] *= down_ratio
        det[:, :4] -= padding
        det[:, :4] /= scale
        classes = det[:, -1]
        per_scale.append({j + 1: det[classes == j, :5].numpy().reshape(-1, 5) for j in range(num_classes)})
    results = {}
    for j in range(1, num_classes + 1):
        results[j] = np.concatenate([d[j] for d in per_scale], axis=0)
        if multi_scale:
            n = soft_nms(results[j], Nt=0.5, method=2)
            results[j] = results[j][:n]
    scores = np.hstack([results[j][:, 4] for j in range(1, num_classes + 1)])
    if len(scores) > max_per_image:
        kth = len(scores) - max_per_image
        thresh = np.partition(scores, kth)[kth]
        for j in range(1, num_classes + 1):
            results[j] = results[j][results[j][:, 4] >= thresh]
    return results


def flip_merge_pose(out, flip_idx):
    """centernet_multi_pose.py:190-211 for B images (rows [B:] mirrored).  `out`: dict of fp32 NCHW head maps [2B,...]."""
    B = out["heatmap"].shape[0] // 2
    r = {"heatmap": flip_merge(out["heatmap"]), "width_height": flip_merge(out["width_height"]), "regression": out["regression"][:B]}
    kp = out["keypoints"]
    _, points, height, width = kp.shape
    fk = kp[B:].flip(-1).reshape(B, points // 2, 2, height, width).clone()
    fk[:, :, 0, :, :] *= -1
    fk = fk[:, flip_idx].reshape(B, points, height, width)
    r["keypoints"] = (kp[:B] + fk) / 2
    fh = out["heatmap_keypoints"][B:].flip(-1)[:, flip_idx]
    r["heatmap_keypoints"] = (out["heatmap_keypoints"][:B] + fh) / 2
    r["heatmap_keypoints_offset"] = out["heatmap_keypoints_offset"][:B]
    return r


def soft_nms_39(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    """utils/nms.py:109-206: as soft_nms, but columns 5..38 travel with the box (swap) while columns 39.. stay in place."""
    N = boxes.shape[0]
    for i in range(N):
        maxscore, maxpos = boxes[i, 4], i
        t = boxes[i, :5].copy()
        pos = i + 1
        while pos < N:
            if maxscore < boxes[pos, 4]:
                maxscore, maxpos = boxes[pos, 4], pos
            pos += 1
        boxes[i, :5] = boxes[maxpos, :5]
        boxes[maxpos, :5] = t
        tmp = boxes[i, 5:39].copy()
        boxes[i, 5:39] = boxes[maxpos, 5:39]
        boxes[maxpos, 5:39] = tmp
        tx1, ty1, tx2, ty2 = (float(v) for v in boxes[i, :4])
        pos = i + 1
        while pos < N:
            x1, y1, x2, y2 = (float(v) for v in boxes[pos, :4])
            area = (x2 - x1 + 1) * (y2 - y1 + 1)
            iw = min(tx2, x2) - max(tx1, x1) + 1
            if iw > 0:
                ih = min(ty2, y2) - max(ty1, y1) + 1
                if ih > 0:
                    ua = float((tx2 - tx1 + 1) * (ty2 - ty1 + 1) + area - iw * ih)
                    ov = iw * ih / ua
                    if method == 1:
                        weight = 1 - ov if ov > Nt else 1
                    elif method == 2:
                        weight = np.exp(-(ov * ov) / sigma)
                    else:
                        weight = 0 if ov > Nt else 1
                    boxes[pos, 4] = weight * float(boxes[pos, 4])
                    if boxes[pos, 4] < threshold:
                        boxes[pos, :5] = boxes[N - 1, :5]
                        tmp = boxes[pos, 5:39].copy()
                        boxes[pos, 5:39] = boxes[N - 1, 5:39]
                        boxes[N - 1, 5:39] = tmp
                        N -= 1
                        pos -= 1
            pos += 1
    return N


def pose_test_step_end(dets, metas, down_ratio=4, max_per_image=20):
    """centernet_multi_pose.py:213-264 for ONE image; dets: list over scales of float32 [K,57]."""
    out = []
    for det, meta in zip(dets, metas):
        det = torch.as_tensor(det, dtype=torch.float32).clone()
        padding = torch.tensor(meta["padding"], dtype=torch.float32)
        scale = torch.tensor(meta["scale"], dtype=torch.float32)
        det[:, :4] *= down_ratio
        det[:, :4] -= torch.cat([padding, padding])
        det[:, :4] /= torch.cat([scale, scale])
        points = det[:, 5:39].view(-1, 17, 2)
        points *= down_ratio
        points -= padding
        points /= scale
        det[:, 5:39] = points.view(-1, 34)
        out.append(det.numpy())
    results = np.concatenate(out, axis=0)
    if len(dets) > 1:
        n = soft_nms_39(results, Nt=0.5, method=2)
        results = results[:n]
    scores = results[:, 4]
    if len(scores) > max_per_image:
        kth = len(scores) - max_per_image
        thresh = np.partition(scores, kth)[kth]
        results = results[results[:, 4] >= thresh]
    return results
