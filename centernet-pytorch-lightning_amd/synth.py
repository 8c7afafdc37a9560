"""Synthetic COCO-like batches for tests and bench (no dataset, no network).

Target tensors follow the layouts the reference's sample encoders emit
(sample/ctdet.py:82-88, sample/multi_pose.py:103-110); the ctdet encoder below re-states
sample/ctdet.py:39-90 + utils/gaussian.py:6-58 in numpy (checked against the imported
reference on its own 2-box fixture, tests/golden/encode_fixture.npz).
"""
import math

import numpy as np
import torch

from . import rng

MEAN = np.array([0.408, 0.447, 0.470], np.float32)   # centernet_detection.py:29
STD = np.array([0.289, 0.274, 0.278], np.float32)    # centernet_detection.py:30

# tests/data/coco_annotation.json of the reference: two "person" boxes (x, y, w, h)
FIXTURE_BOXES = [([412.8, 157.61, 53.05, 138.01], 0), ([384.43, 172.21, 15.12, 35.74], 0)]


def gaussian_radius(h, w, min_overlap=0.7):
    """utils/gaussian.py:6-26."""
    b1 = h + w
    c1 = w * h * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 + math.sqrt(b1 ** 2 - 4 * c1)) / 2
    b2 = 2 * (h + w)
    c2 = (1 - min_overlap) * w * h
    r2 = (b2 + math.sqrt(b2 ** 2 - 16 * c2)) / 2
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (h + w)
    c3 = (min_overlap - 1) * w * h
    r3 = (b3 + math.sqrt(b3 ** 2 - 4 * a3 * c3)) / 2
    return min(r1, r2, r3)


def _gauss2d(diam, sigma):
    """utils/gaussian.py:29-38 (float32 like the torch original)."""
    m = (diam - 1.0) / 2.0
    a = np.arange(-m, m + 1, dtype=np.float32)
    g = np.exp(-(a[None, :] * a[None, :] + a[:, None] * a[:, None]) / np.float32(2 * sigma * sigma)).astype(np.float32)
    g[g < np.finfo(np.float32).eps * g.max()] = 0
    return g


def splat_umich(hm, cx, cy, radius):
    """utils/gaussian.py:41-58: element-wise max of the map with a truncated gaussian."""
    g = _gauss2d(2 * radius + 1, (2 * radius + 1) / 6)
    H, W = hm.shape
    l, r = min(cx, radius), min(W - cx, radius + 1)
    t, b = min(cy, radius), min(H - cy, radius + 1)
    if r + l > 0 and t + b > 0:
        sub = hm[cy - t:cy + b, cx - l:cx + r]
        np.maximum(sub, g[radius - t:radius + b, radius - l:radius + r], out=sub)


def encode_ctdet(boxes, in_h=512, in_w=512, down=4, num_classes=80, max_objs=128, gaussian="umich"):
    """boxes: list of ([x,y,w,h] in input pixels, class_id). Returns dict of numpy arrays.  gaussian "msra": the reference's other
    option (sample/ctdet.py:53-55) — draw_msra_gaussian with the INTEGER radius as sigma (radius 0 is its 0/0 -> NaN pixel)."""
    oh, ow = in_h // down, in_w // down
    hm = np.zeros((num_classes, oh, ow), np.float32)
    wh = np.zeros((max_objs, 2), np.float32)
    reg = np.zeros((max_objs, 2), np.float32)
    msk = np.zeros((max_objs,), bool)
    ind = np.zeros((max_objs,), np.int64)
    for k, (bb, cls) in enumerate(boxes[:max_objs]):
        box = np.array([bb[0], bb[1], bb[0] + bb[2], bb[1] + bb[3]], np.float32) / np.float32(down)
        box[0::2] = np.clip(box[0::2], 0, ow - 1)
        box[1::2] = np.clip(box[1::2], 0, oh - 1)
        h, w = box[3] - box[1], box[2] - box[0]
        if h > 0 and w > 0:
            rad = max(0, int(gaussian_radius(math.ceil(h), math.ceil(w))))
            ct = np.array([(box[0] + box[2]) / 2, (box[1] + box[3]) / 2], np.float32)
            ci = ct.astype(np.int32)
            if gaussian == "msra":
                with np.errstate(invalid="ignore", divide="ignore"):
                    splat_msra(hm[cls], int(ci[0]), int(ci[1]), rad)
            else:
                splat_umich(hm[cls], int(ci[0]), int(ci[1]), rad)
            wh[k] = (w, h)
            ind[k] = int(ci[1]) * ow + int(ci[0])
            reg[k] = ct - ci
            msk[k] = True
    return {"heatmap": hm, "regression_mask": msk, "indices": ind, "width_height": wh, "regression": reg}


def splat_msra(hm, cx, cy, sigma):
    """utils/gaussian.py:61-83 (draw_msra_gaussian) as called from sample/multi_pose.py:100 — `sigma` is the FLOAT
    gaussian_radius (not truncated), the centre is the integer keypoint.  Quirks kept: a gaussian whose 3-sigma box touches
    the border is dropped entirely (:67-68, with the reference's swapped w/h names); the box corners truncate toward zero;
    the peak sits at `ul + floor(3 sigma + 0.5)`, which is one pixel left/up of the keypoint when frac(3 sigma) < 0.5."""
    t = sigma * 3
    H, W = hm.shape
    ul = (int(cx - t), int(cy - t))
    br = (int(cx + t + 1), int(cy + t + 1))
    if br[0] >= W or br[1] >= H or ul[0] < 0 or ul[1] < 0:
        return
    size = 2 * t + 1
    x = np.arange(0, size, 1, np.float32)
    y = x[:, None]
    x0 = y0 = size // 2
    g = np.exp(-((x - np.float32(x0)) ** 2 + (y - np.float32(y0)) ** 2) / np.float32(2 * sigma ** 2))
    sub = hm[ul[1]:br[1], ul[0]:br[0]]
    np.maximum(sub, g[:br[1] - ul[1], :br[0] - ul[0]], out=sub)


def encode_multi_pose(anns, in_h=512, in_w=512, down=4, joints=17, max_objs=128):
    """sample/multi_pose.py:35-112 in numpy.  anns: list of ([x,y,w,h], keypoints [joints*3] as x,y,visibility) in input pixels.
    Returns the six pose target arrays (the ctdet part of a multi_pose sample is `encode_ctdet(..., num_classes=1)`)."""
    oh, ow = in_h // down, in_w // down
    T = {"heatmap_keypoints": np.zeros((joints, oh, ow), np.float32),
         "keypoints": np.zeros((max_objs, 2 * joints), np.float32),
         "keypoints_mask": np.zeros((max_objs, 2 * joints), bool),
         "heatmap_keypoints_offset": np.zeros((max_objs * joints, 2), np.float32),
         "heatmap_keypoints_indices": np.zeros((max_objs * joints,), np.int64),
         "heatmap_keypoints_mask": np.zeros((max_objs * joints,), bool)}
    f32 = np.float32
    for k, (bb, kps) in enumerate(anns[:max_objs]):
        box = np.array([bb[0], bb[1], bb[0] + bb[2], bb[1] + bb[3]], f32) / f32(down)
        box[0::2] = np.clip(box[0::2], 0, ow - 1)
        box[1::2] = np.clip(box[1::2], 0, oh - 1)
        ctx, cty = int((box[0] + box[2]) / f32(2)), int((box[1] + box[3]) / f32(2))     # torch.IntTensor([...]): truncation
        h, w = box[3] - box[1], box[2] - box[0]
        if not (h > 0 and w > 0):
            continue
        sigma = gaussian_radius(math.ceil(h), math.ceil(w))
        pts = np.array(kps, f32).reshape(joints, 3)
        for j in range(joints):
            if pts[j, 2] == 0:
                continue
            px = f32(np.clip(pts[j, 0] / f32(down), 0, ow - 1))
            py = f32(np.clip(pts[j, 1] / f32(down), 0, oh - 1))
            T["keypoints"][k, 2 * j:2 * j + 2] = (px - f32(ctx), py - f32(cty))
            T["keypoints_mask"][k, 2 * j:2 * j + 2] = True
            ix, iy = int(px), int(py)
            T["heatmap_keypoints_offset"][k * joints + j] = (px - f32(ix), py - f32(iy))
            T["heatmap_keypoints_indices"][k * joints + j] = iy * ow + ix
            T["heatmap_keypoints_mask"][k * joints + j] = True
            splat_msra(T["heatmap_keypoints"][j], ix, iy, sigma)
    return T


def random_pose_anns(seed, img_idx, in_h=512, in_w=512, max_n=8, joints=17):
    """person boxes + keypoints scattered in and slightly around each box, ~25 % invisible (v = 0)."""
    anns = []
    for k, (bb, _) in enumerate(random_boxes(seed, img_idx, in_h, in_w, 1, max_n)):
        u = rng.uniform(seed, f"pkp{img_idx}_{k}", (joints, 3))
        kps = []
        for j in range(joints):
            kps += [float(np.float32(bb[0] + (1.3 * u[j, 0] - 0.15) * bb[2])), float(np.float32(bb[1] + (1.3 * u[j, 1] - 0.15) * bb[3])),
                    0.0 if u[j, 2] < 0.25 else (1.0 if u[j, 2] < 0.5 else 2.0)]
        anns.append(([float(np.float32(v)) for v in bb], kps))
    return anns


def random_boxes(seed, img_idx, in_h=512, in_w=512, num_classes=80, max_n=20):
    n = int(rng.randint(seed, f"nobj{img_idx}", (1,), 1, max_n + 1)[0])
    wh = rng.uniform(seed, f"bwh{img_idx}", (n, 2), 8.0, 256.0)
    xy = rng.uniform(seed, f"bxy{img_idx}", (n, 2), 0.0, 1.0)
    cls = rng.randint(seed, f"bcls{img_idx}", (n,), 0, num_classes)
    out = []
    for i in range(n):
        w = min(float(wh[i, 0]), in_w - 1.0)
        h = min(float(wh[i, 1]), in_h - 1.0)
        x = float(xy[i, 0]) * (in_w - 1 - w)
        y = float(xy[i, 1]) * (in_h - 1 - h)
        out.append(([x, y, w, h], int(cls[i])))
    return out


def images(seed, batch, in_h=512, in_w=512, start=0):
    """[B,3,H,W] fp32: U[0,1) normalised with the reference mean/std (tests/utilities.py:19 +
    centernet_detection.py:151)."""
    x = np.stack([rng.uniform(seed, f"img{start + i}", (3, in_h, in_w)) for i in range(batch)])
    return torch.from_numpy((x - MEAN[None, :, None, None]) / STD[None, :, None, None])


def ctdet_batch(seed, batch, in_h=512, in_w=512, num_classes=80, start=0, fixture=False):
    """(images, target-dict) like a collated CenterDetectionSample batch."""
    enc = [encode_ctdet(FIXTURE_BOXES if fixture else random_boxes(seed, start + i, in_h, in_w, num_classes),
                        in_h, in_w, 4, num_classes) for i in range(batch)]
    tgt = {k: torch.from_numpy(np.stack([e[k] for e in enc])) for k in enc[0]}
    return images(seed, batch, in_h, in_w, start), tgt


def pose_batch(seed, batch, in_h=512, in_w=512, start=0, joints=17, max_objs=128):
    """Synthetic multi_pose targets with the layouts of sample/multi_pose.py:103-110
    (values are synthetic: person boxes + uniformly placed visible keypoints inside them)."""
    oh, ow = in_h // 4, in_w // 4
    imgs = images(seed, batch, in_h, in_w, start)
    T = {"heatmap": np.zeros((batch, 1, oh, ow), np.float32),
         "regression_mask": np.zeros((batch, max_objs), bool),
         "indices": np.zeros((batch, max_objs), np.int64),
         "width_height": np.zeros((batch, max_objs, 2), np.float32),
         "regression": np.zeros((batch, max_objs, 2), np.float32),
         "heatmap_keypoints": np.zeros((batch, joints, oh, ow), np.float32),
         "keypoints": np.zeros((batch, max_objs, 2 * joints), np.float32),
         "keypoints_mask": np.zeros((batch, max_objs, 2 * joints), bool),
         "heatmap_keypoints_offset": np.zeros((batch, max_objs * joints, 2), np.float32),
         "heatmap_keypoints_indices": np.zeros((batch, max_objs * joints), np.int64),
         "heatmap_keypoints_mask": np.zeros((batch, max_objs * joints), bool)}
    for b in range(batch):
        boxes = [(bb, 0) for bb, _ in random_boxes(seed, start + b, in_h, in_w, 1, 8)]
        e = encode_ctdet(boxes, in_h, in_w, 4, 1, max_objs)
        for k in ("heatmap", "regression_mask", "indices", "width_height", "regression"):
            T[k][b] = e[k]
        for k, (bb, _) in enumerate(boxes):
            if not e["regression_mask"][k]:
                continue
            cy, cx = divmod(int(e["indices"][k]), ow)
            rad = max(0, int(gaussian_radius(math.ceil(bb[3] / 4), math.ceil(bb[2] / 4))))
            u = rng.uniform(seed, f"kp{start + b}_{k}", (joints, 3))
            for j in range(joints):
                if u[j, 2] < 0.3:
                    continue
                px = np.float32((bb[0] + u[j, 0] * bb[2]) / 4)
                py = np.float32((bb[1] + u[j, 1] * bb[3]) / 4)
                ix, iy = int(px), int(py)
                if not (0 <= ix < ow and 0 <= iy < oh):
                    continue
                T["keypoints"][b, k, 2 * j:2 * j + 2] = (px - cx, py - cy)
                T["keypoints_mask"][b, k, 2 * j:2 * j + 2] = True
                T["heatmap_keypoints_offset"][b, k * joints + j] = (px - ix, py - iy)
                T["heatmap_keypoints_indices"][b, k * joints + j] = iy * ow + ix
                T["heatmap_keypoints_mask"][b, k * joints + j] = True
                splat_umich(T["heatmap_keypoints"][b, j], ix, iy, rad)
    return imgs, {k: torch.from_numpy(v) for k, v in T.items()}


def tta_head_maps(seed, channels, sizes):
    """Seeded raw head maps for the test scales: {name: [1, C, H, W]} per scale; the heat-map logits are biased so that the
    decode sees a few hundred separated peaks."""
    outs = []
    for i, (h, w) in enumerate(sizes):
        o = {}
        for name, (c, kind) in channels.items():
            if kind == "logit":
                o[name] = rng.t_normal(seed, f"{name}{i}", (1, c, h, w)) * 2.0 - 2.0
            elif kind == "size":
                o[name] = rng.t_uniform(seed, f"{name}{i}", (1, c, h, w), 2.0, 12.0)
            elif kind == "kps":
                o[name] = rng.t_normal(seed, f"{name}{i}", (1, c, h, w), 0, 4.0)
            else:
                o[name] = rng.t_uniform(seed, f"{name}{i}", (1, c, h, w))
        outs.append(o)
    return outs


DET_MAPS = {"heatmap": (3, "logit"), "width_height": (2, "size"), "regression": (2, "off")}
POSE_MAPS = {"heatmap": (1, "logit"), "width_height": (2, "size"), "regression": (2, "off"), "keypoints": (34, "kps"),
             "heatmap_keypoints": (17, "logit"), "heatmap_keypoints_offset": (2, "off")}
# 128x128 image at scales 1 and 0.75: padded to (size | 31) + 1 = 160 / 128 -> 40x40 and 32x32 maps
TTA_SIZES = [(40, 40), (32, 32)]
TTA_METAS = [{"scale": [1.0, 1.0], "padding": [16, 16]}, {"scale": [0.75, 0.75], "padding": [16, 16]}]
