"""multi_pose ground-truth encoding on the device (reference: CenterNet/sample/multi_pose.py:35-112).

The reference walks objects × joints in Python per sample and splats one numpy gaussian per visible keypoint.  Here the
whole batch is ONE launch (`cn_encode_multi_pose`, a workgroup per object-joint); together with `encode_ctdet_batch(...,
num_classes=1)` it yields the full multi_pose target dict in HBM.  `MultiPoseSample` keeps the reference's transform
signature (and its default "msra" gaussian — the only one the reference's training script uses).
"""
import torch

from .._hip import call
from .ctdet import encode_ctdet_batch


def encode_multi_pose_batch(boxes, keypoints, counts, in_h, in_w, down_ratio=4, with_ctdet=True):
    """boxes fp32 [B, M, 4] (x, y, w, h), keypoints fp32 [B, M, J, 3] (x, y, visibility), counts int32 [B] — CUDA tensors.
    Returns the collated pose targets (sample/multi_pose.py:103-110) and, `with_ctdet`, the 1-class ctdet targets too
    (the reference composes CenterDetectionSample(num_classes=1) with MultiPoseSample)."""
    assert boxes.is_cuda, "the encoder has no host path: move the annotations to the GPU"
    B, M, _ = boxes.shape
    J = keypoints.shape[2]
    oh, ow = in_h // down_ratio, in_w // down_ratio
    dev = boxes.device
    boxes, counts = boxes.float().contiguous(), counts.int().contiguous()
    hm = torch.zeros((B, J, oh, ow), dtype=torch.float32, device=dev)
    kp = torch.empty((B, M, 2 * J), dtype=torch.float32, device=dev)
    kpm = torch.empty((B, M, 2 * J), dtype=torch.uint8, device=dev)
    off = torch.empty((B, M * J, 2), dtype=torch.float32, device=dev)
    ind = torch.empty((B, M * J), dtype=torch.int64, device=dev)
    hpm = torch.empty((B, M * J), dtype=torch.uint8, device=dev)
    call("cn_encode_multi_pose", boxes, keypoints.float().contiguous(), counts, hm, kp, kpm, off, ind, hpm, B, M, J, oh, ow,
         int(down_ratio))
    out = {"heatmap_keypoints": hm, "keypoints": kp, "keypoints_mask": kpm.bool(), "heatmap_keypoints_offset": off,
           "heatmap_keypoints_indices": ind, "heatmap_keypoints_mask": hpm.bool()}
    if with_ctdet:
        cls = torch.zeros((B, M), dtype=torch.int32, device=dev)
        out.update(encode_ctdet_batch(boxes, cls, counts, in_h, in_w, down_ratio, num_classes=1))
    return out


class MultiPoseSample:
    """sample/multi_pose.py:9-112: `(img, annotations) -> (img, pose targets)`; img is a CUDA [3, H, W] tensor here."""

    def __init__(self, down_ratio=4, max_objects=128, gaussian_type="msra", num_joints=17):
        if gaussian_type != "msra":
            raise NotImplementedError("only the default msra gaussian (sample/multi_pose.py:11, 60-62): with 'umich' the reference "
                                      "passes the FLOAT radius to draw_umich_gaussian, which slices with it (TypeError) — "
                                      "there is no reference behaviour to reproduce")
        self.down_ratio, self.max_objects, self.num_joints = down_ratio, max_objects, num_joints

    def __call__(self, img, target):
        _, in_h, in_w = img.shape
        n = min(len(target), self.max_objects)
        boxes = torch.zeros((1, self.max_objects, 4), dtype=torch.float32)
        kps = torch.zeros((1, self.max_objects, self.num_joints, 3), dtype=torch.float32)
        for k in range(n):
            boxes[0, k] = torch.tensor(target[k]["bbox"], dtype=torch.float32)
            kps[0, k] = torch.tensor(target[k]["keypoints"], dtype=torch.float32).view(self.num_joints, 3)
        t = encode_multi_pose_batch(boxes.to(img.device), kps.to(img.device), torch.tensor([n], dtype=torch.int32, device=img.device),
                                    in_h, in_w, self.down_ratio, with_ctdet=False)
        return img, {k: v[0] for k, v in t.items()}
