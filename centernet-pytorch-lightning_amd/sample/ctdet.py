"""ctdet ground-truth encoding on the device (reference: CenterNet/sample/ctdet.py:39-90, utils/gaussian.py).

The reference encodes one sample at a time in Python on the host (gaussian splat per object).  Here a whole batch of
raw COCO boxes is encoded by ONE launch (`cn_encode_ctdet`): the loader only has to collate boxes, and the targets are
born in HBM where the loss reads them.  `CenterDetectionSample` keeps the reference's transform signature.
"""
import torch

from .._hip import call


def encode_ctdet_batch(boxes, classes, counts, in_h, in_w, down_ratio=4, num_classes=80, gaussian_type="umich"):
    """boxes fp32 [B, M, 4] (x, y, w, h in input pixels), classes int32 [B, M], counts int32 [B] — CUDA tensors.
    Returns the target dict of a collated CenterDetectionSample batch (sample/ctdet.py:80-88)."""
    assert boxes.is_cuda, "the encoder has no host path: move the boxes to the GPU"
    B, M, _ = boxes.shape
    oh, ow = in_h // down_ratio, in_w // down_ratio
    dev = boxes.device
    heat = torch.zeros((B, num_classes, oh, ow), dtype=torch.float32, device=dev)
    mask = torch.empty((B, M), dtype=torch.uint8, device=dev)
    ind = torch.empty((B, M), dtype=torch.int64, device=dev)
    wh = torch.empty((B, M, 2), dtype=torch.float32, device=dev)
    reg = torch.empty((B, M, 2), dtype=torch.float32, device=dev)
    call("cn_encode_ctdet", boxes.float().contiguous(), classes.int().contiguous(), counts.int().contiguous(), heat, mask, ind, wh,
         reg, B, M, num_classes, oh, ow, int(down_ratio), {"umich": 0, "msra": 1}[gaussian_type])
    return {"heatmap": heat, "regression_mask": mask.bool(), "indices": ind, "width_height": wh, "regression": reg}


class CenterDetectionSample:
    """sample/ctdet.py:10-90: `(img, annotations) -> (img, targets)`; img is a CUDA [3, H, W] tensor here."""

    def __init__(self, down_ratio=4, num_classes=80, max_objects=128, gaussian_type="umich"):
        if gaussian_type not in ("umich", "msra"):
            raise ValueError(f"gaussian_type {gaussian_type!r}: 'umich' or 'msra' (sample/ctdet.py:53-55)")
        self.down_ratio, self.num_classes, self.max_objects, self.gaussian_type = down_ratio, num_classes, max_objects, gaussian_type

    def __call__(self, img, target):
        _, in_h, in_w = img.shape            # the reference unpacks (_, input_w, input_h); square inputs in every config
        n = min(len(target), self.max_objects)
        boxes = torch.zeros((1, self.max_objects, 4), dtype=torch.float32)
        cls = torch.zeros((1, self.max_objects), dtype=torch.int32)
        for k in range(n):
            a = target[k]
            boxes[0, k] = torch.tensor(a["bbox"], dtype=torch.float32)
            cls[0, k] = a["class_id"] if "class_id" in a else int(a["category_id"]) - 1
        t = encode_ctdet_batch(boxes.to(img.device), cls.to(img.device), torch.tensor([n], dtype=torch.int32, device=img.device),
                               in_h, in_w, self.down_ratio, self.num_classes, self.gaussian_type)
        return img, {k: v[0] for k, v in t.items()}
