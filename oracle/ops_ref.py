"""ORACLE (test infrastructure): decode primitives, losses, ctdet / multi_pose decode.

Plain torch CPU ops.  Tie rule: wherever the reference calls ``torch.topk`` (tie order
unspecified) this restatement uses a *stable* descending order — equal scores keep
ascending flat index — which is the rule the HIP kernels implement.  On tie-free inputs it
is bit-identical to the reference (checked against tests/golden/decode_*.npz).
"""
import torch
import torch.nn.functional as F


def peak_mask(heat):
    """utils/decode.py:5-9 — 3x3/s1/p1 max-pool, keep where pooled == centre (plateaus keep all)."""
    pooled = F.max_pool2d(heat, 3, stride=1, padding=1)
    return pooled == heat


def nms(heat, kernel=3):
    """utils/decode.py:5-10 (any odd kernel; the reference itself only ever passes the default 3)."""
    if kernel == 3:
        return heat * peak_mask(heat).float()
    pooled = F.max_pool2d(heat, kernel, stride=1, padding=(kernel - 1) // 2)
    return heat * (pooled == heat).float()


def stable_topk(x, k):
    """Descending top-k along the last dim; ties -> lower index first."""
    order = torch.argsort(x, dim=-1, descending=True, stable=True)[..., :k]
    return torch.gather(x, -1, order), order


def topk_channel(scores, K):
    """utils/decode.py:31-40 — per (batch, class) top-K over H*W."""
    b, c, h, w = scores.shape
    s, i = stable_topk(scores.reshape(b, c, h * w), K)
    ys = torch.div(i, w, rounding_mode="floor").float()
    xs = (i % w).float()
    return s, i, ys, xs


def topk(scores, K):
    """utils/decode.py:13-28 — per-class top-K, then top-K over C*K; returns
    (score, flat index y*W+x, class, y, x) each [B,K]."""
    b, c, h, w = scores.shape
    s1, i1, ys1, xs1 = topk_channel(scores, K)
    s2, j = stable_topk(s1.reshape(b, c * K), K)
    cls = torch.div(j, K, rounding_mode="floor").int()
    pick = lambda t: torch.gather(t.reshape(b, c * K), 1, j)
    return s2, pick(i1), cls, pick(ys1), pick(xs1)


def gather_feat(feat, ind):
    """utils/decode.py:48-56 (mask=None branch): feat [B,HW,C], ind [B,N] -> [B,N,C]."""
    return torch.gather(feat, 1, ind.unsqueeze(2).expand(-1, -1, feat.size(2)))


def transpose_and_gather_feat(feat, ind):
    """utils/decode.py:59-63: NCHW map -> rows at flat spatial indices."""
    b, c = feat.shape[:2]
    return gather_feat(feat.permute(0, 2, 3, 1).reshape(b, -1, c), ind)


def sigmoid_clamped(x, clamp=1e-4):
    """utils/decode.py:43-45 (sigmoid is in place on x in the reference; here functional)."""
    return torch.clamp(torch.sigmoid(x), min=clamp, max=1 - clamp)


def focal_loss(pred, gt):
    """utils/losses.py:14-39 penalty-reduced focal loss; pred already sigmoid-clamped."""
    pos = (gt == 1).float()
    neg = (gt < 1).float()
    pos_term = (torch.log(pred) * (1 - pred) ** 2 * pos).sum()
    neg_term = (torch.log(1 - pred) * pred ** 2 * (1 - gt) ** 4 * neg).sum()
    n = pos.sum()
    if n == 0:
        return -neg_term
    return -(pos_term + neg_term) / n


def reg_l1_loss(output, mask, ind, target):
    """utils/losses.py:53-63 — mask [B,N] bool, expanded over channels."""
    pred = transpose_and_gather_feat(output, ind)
    m = mask.unsqueeze(2).expand_as(pred).float()
    return F.l1_loss(pred * m, target * m, reduction="sum") / (m.sum() + 1e-4)


def norm_reg_l1_loss(output, mask, ind, target):
    """utils/losses.py:66-78 — L1 between pred / (target + 1e-4) and 1, mask [B,N] bool expanded over channels."""
    pred = transpose_and_gather_feat(output, ind)
    m = mask.unsqueeze(2).expand_as(pred).float()
    pred = pred / (target + 1e-4)
    return F.l1_loss(pred * m, (target * 0 + 1) * m, reduction="sum") / (m.sum() + 1e-4)


def reg_weighted_l1_loss(output, mask, ind, target):
    """utils/losses.py:81-91 — mask already [B,N,C]."""
    pred = transpose_and_gather_feat(output, ind)
    m = mask.float()
    return F.l1_loss(pred * m, target * m, reduction="sum") / (m.sum() + 1e-4)


def ctdet_loss(out, target, hm_weight=1.0, wh_weight=0.1, off_weight=1.0):
    """centernet_detection.py:97-130 for one stack; ``out`` holds raw head outputs."""
    hm = sigmoid_clamped(out["heatmap"])
    hm_loss = focal_loss(hm, target["heatmap"])
    wh_loss = reg_l1_loss(out["width_height"], target["regression_mask"], target["indices"], target["width_height"])
    off_loss = reg_l1_loss(out["regression"], target["regression_mask"], target["indices"], target["regression"])
    loss = hm_weight * hm_loss + wh_weight * wh_loss + off_weight * off_loss
    return loss, {"loss": loss, "hm_loss": hm_loss, "wh_loss": wh_loss, "off_loss": off_loss}


def multi_pose_loss(out, target, hm_weight=1.0, wh_weight=0.1, off_weight=1.0, hp_weight=1.0, hm_hp_weight=1.0):
    """centernet_multi_pose.py:97-155 for one stack."""
    hm = sigmoid_clamped(out["heatmap"])
    hm_hp = sigmoid_clamped(out["heatmap_keypoints"])
    hm_loss = focal_loss(hm, target["heatmap"])
    wh_loss = reg_l1_loss(out["width_height"], target["regression_mask"], target["indices"], target["width_height"])
    off_loss = reg_l1_loss(out["regression"], target["regression_mask"], target["indices"], target["regression"])
    kp_loss = reg_weighted_l1_loss(out["keypoints"], target["keypoints_mask"], target["indices"], target["keypoints"])
    hm_kp_loss = focal_loss(hm_hp, target["heatmap_keypoints"])
    hm_off_loss = reg_l1_loss(out["heatmap_keypoints_offset"], target["heatmap_keypoints_mask"],
                              target["heatmap_keypoints_indices"], target["heatmap_keypoints_offset"])
    loss = (hm_weight * hm_loss + wh_weight * wh_loss + off_weight * off_loss
            + hp_weight * kp_loss + hm_hp_weight * hm_kp_loss + off_weight * hm_off_loss)
    return loss, {"loss": loss, "hm_loss": hm_loss, "kp_loss": kp_loss, "hm_kp_loss": hm_kp_loss,
                  "hm_offset_loss": hm_off_loss, "wh_loss": wh_loss, "off_loss": off_loss}


def _boxes(xs, ys, wh):
    return torch.cat([xs - wh[..., 0:1] / 2, ys - wh[..., 1:2] / 2,
                      xs + wh[..., 0:1] / 2, ys + wh[..., 1:2] / 2], dim=2)


def ctdet_decode(heat, wh, reg=None, K=100, return_aux=False):
    """decode/ctdet.py:6-38 -> [B,K,6] = x1,y1,x2,y2,score,class."""
    b = heat.size(0)
    scores, inds, clses, ys, xs = topk(nms(heat), K)
    if reg is not None:
        r = transpose_and_gather_feat(reg, inds)
        xs = xs.view(b, K, 1) + r[:, :, 0:1]
        ys = ys.view(b, K, 1) + r[:, :, 1:2]
    else:
        xs = xs.view(b, K, 1) + 0.5
        ys = ys.view(b, K, 1) + 0.5
    w = transpose_and_gather_feat(wh, inds)
    det = torch.cat([_boxes(xs, ys, w), scores.view(b, K, 1), clses.view(b, K, 1).float()], dim=2)
    return (det, inds, clses) if return_aux else det


def multi_pose_decode(heat, wh, kps, reg=None, hm_hp=None, hp_offset=None, K=100):
    """decode/multi_pose.py:7-96 -> [B,K,4+1+2J+1+J].  ``dist.min`` ties -> first index."""
    b = heat.size(0)
    J = kps.shape[1] // 2
    scores, inds, clses, ys, xs = topk(nms(heat), K)
    kp = transpose_and_gather_feat(kps, inds).clone()
    kp[..., 0::2] += xs.view(b, K, 1)
    kp[..., 1::2] += ys.view(b, K, 1)
    if reg is not None:
        r = transpose_and_gather_feat(reg, inds)
        xs = xs.view(b, K, 1) + r[:, :, 0:1]
        ys = ys.view(b, K, 1) + r[:, :, 1:2]
    else:
        xs = xs.view(b, K, 1) + 0.5
        ys = ys.view(b, K, 1) + 0.5
    w = transpose_and_gather_feat(wh, inds)
    bboxes = _boxes(xs, ys, w)
    hm_score = None
    if hm_hp is not None:
        thresh = 0.1
        kp = kp.view(b, K, J, 2).permute(0, 2, 1, 3).contiguous()          # B,J,K,2
        s, hi, hy, hx = topk_channel(nms(hm_hp), K)                         # B,J,K
        if hp_offset is not None:
            o = transpose_and_gather_feat(hp_offset, hi.reshape(b, -1)).view(b, J, K, 2)
            hx = hx + o[..., 0]
            hy = hy + o[..., 1]
        else:
            hx = hx + 0.5
            hy = hy + 0.5
        m = (s > thresh).float()
        s = (1 - m) * -1 + m * s
        hy = (1 - m) * (-10000) + m * hy
        hx = (1 - m) * (-10000) + m * hx
        cand = torch.stack([hx, hy], dim=-1)                                # B,J,K,2
        d = ((kp.unsqueeze(3) - cand.unsqueeze(2)) ** 2).sum(dim=4) ** 0.5  # B,J,K(reg),K(cand)
        min_d, min_i = d.min(dim=3)
        s = s.gather(2, min_i).unsqueeze(-1)
        min_d = min_d.unsqueeze(-1)
        sel = cand.gather(2, min_i.unsqueeze(-1).expand(-1, -1, -1, 2))     # B,J,K,2
        l = bboxes[:, :, 0].view(b, 1, K, 1)
        t = bboxes[:, :, 1].view(b, 1, K, 1)
        r_ = bboxes[:, :, 2].view(b, 1, K, 1)
        bt = bboxes[:, :, 3].view(b, 1, K, 1)
        bad = ((sel[..., 0:1] < l) | (sel[..., 0:1] > r_) | (sel[..., 1:2] < t) | (sel[..., 1:2] > bt)
               | (s < thresh) | (min_d > torch.max(bt - t, r_ - l) * 0.3)).float()
        s = s * (1 - bad)
        hm_score = s.view(b, K, J)   # the reference *reshapes* [B,J,K,1] -> [B,K,J] (no permute): multi_pose.py:90
        kp = (1 - bad) * sel + bad * kp
        kp = kp.permute(0, 2, 1, 3).contiguous().view(b, K, 2 * J)
    parts = [bboxes, scores.view(b, K, 1), kp, clses.view(b, K, 1).float()]
    if hm_score is not None:
        parts.append(hm_score)
    return torch.cat(parts, dim=2)
