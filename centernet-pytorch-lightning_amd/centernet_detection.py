"""`CenterNetDetection` (reference: CenterNet/centernet_detection.py:28-265) — forward / loss / decode / test step on the HIP
path.  COCO evaluation (pycocotools) and the CLI are outside the hot-path scope."""
import torch

from .centernet import CenterNet
from .decode.ctdet import ctdet_decode
from .models.heads import CenterHead
from .utils.decode import sigmoid_clamped
from .utils.losses import FocalLoss, RegL1Loss
from .utils import post


class CenterNetDetection(CenterNet):
    mean = [0.408, 0.447, 0.470]
    std = [0.289, 0.274, 0.278]
    max_objs = 128
    valid_ids = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 27, 28, 31, 32, 33,
                 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61,
                 62, 63, 64, 65, 67, 70, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 84, 85, 86, 87, 88, 89, 90]

    def __init__(self, arch, learning_rate=1e-4, learning_rate_milestones=None, hm_weight=1, wh_weight=0.1, off_weight=1,
                 num_classes=80, test_coco=None, test_coco_ids=None, test_scales=None, test_flip=False,
                 compute_dtype=torch.bfloat16):
        super().__init__(arch, compute_dtype=compute_dtype)
        self.num_classes = num_classes
        heads = {"heatmap": self.num_classes, "width_height": 2, "regression": 2}
        self.heads = torch.nn.ModuleList([CenterHead(heads, self.backbone.out_channels, self.head_conv)
                                          for _ in range(self.num_stacks)])
        self.learning_rate_milestones = learning_rate_milestones if learning_rate_milestones is not None else []
        self.test_coco, self.test_coco_ids = test_coco, test_coco_ids
        self.test_max_per_image = 100
        self.test_scales = [1] if test_scales is None else test_scales
        self.test_flip = test_flip
        self.criterion = FocalLoss()
        self.criterion_regression = RegL1Loss()
        self.criterion_width_height = RegL1Loss()
        self.save_hyperparameters()

    def forward(self, x):
        return [head(out) for head, out in zip(self.heads, self.backbone(x))]

    def loss(self, outputs, target):
        hm, wh, off = [], [], []
        num_stacks = len(outputs)
        for output in outputs:
            output["heatmap"], t = self._sigmoid_focal(self.criterion, output["heatmap"], target["heatmap"])
            hm.append(t)
            wh.append(self.criterion_width_height(output["width_height"], target["regression_mask"], target["indices"],
                                                  target["width_height"]))
            off.append(self.criterion_regression(output["regression"], target["regression_mask"], target["indices"],
                                                 target["regression"]))
        # centernet_detection.py:108-116: per-term sums over the stacks, then the weighted total / num_stacks — one launch
        # (ops.weighted_sum) instead of a dozen scalar ATen kernels; with one stack a term's sum IS the term
        h = self.hparams
        hm_loss, wh_loss, off_loss = (ts[0] if num_stacks == 1 else self._sum_terms(ts) for ts in (hm, wh, off))
        loss = self._weighted_total([hm, wh, off], [h.hm_weight, h.wh_weight, h.off_weight], num_stacks)
        return loss, {"loss": loss, "hm_loss": hm_loss, "wh_loss": wh_loss, "off_loss": off_loss}

    @torch.no_grad()
    def decode(self, output, K=100, fused=False):
        """The decode call of test_step_end (centernet_detection.py:183-187): sigmoid in place, then ctdet_decode.
        fused=True (opt-in, throughput path): the sigmoid is applied by the top-K kernel on load (cn_ctdet_decode_logits, no pass over
        the map, `output["heatmap"]` keeps the logits); its expf-based sigmoid may differ from ATen's in the last bit of a score."""
        if fused:
            return ctdet_decode(output["heatmap"], output["width_height"], reg=output["regression"], K=K, logits_clamp=0.0)
        return ctdet_decode(output["heatmap"].sigmoid_(), output["width_height"], reg=output["regression"], K=K)

    @torch.no_grad()
    def test_step(self, batch, batch_idx):
        """centernet_detection.py:132-173 for a BATCH of images in [0, 1] (the reference runs batch size 1): per test scale
        resize -> zero-pad to `(size | padding) + 1` -> normalise -> (+ mirrored copy) in one launch, forward, and the mirrored
        head maps folded back (heat map and sizes averaged, offsets from the unflipped pass)."""
        img, _ = batch
        B = img.shape[0]
        image_id = ([self.test_coco_ids[batch_idx * B + i] for i in range(B)] if self.test_coco_ids
                    else [batch_idx * B + i for i in range(B)])
        outputs, meta = [], []
        for scale in self.test_scales:
            _, _, height, width = img.shape
            nh, nw = int(height * scale), int(width * scale)
            pad_y, pad_x = post.tta_pad(nh, self.padding), post.tta_pad(nw, self.padding)
            x = post.tta_prepare_scaled(img, nh, nw, self.mean, self.std, pad_x, pad_y, self.test_flip)   # resize in the same launch
            out = self(x)[-1]
            if self.test_flip:
                out = {"heatmap": post.flip_merge(out["heatmap"]), "width_height": post.flip_merge(out["width_height"]),
                       "regression": out["regression"][:B].contiguous()}
            outputs.append(out)
            meta.append({"scale": [nw / width, nh / height], "padding": [pad_x, pad_y]})
        return image_id, outputs, meta

    @torch.no_grad()
    def test_step_end(self, outputs):
        """centernet_detection.py:173-225 for the batch: decode every scale, then ONE launch maps the boxes back to the
        image, groups them by class, merges the scales with soft-NMS and keeps the best `test_max_per_image`; one host copy.
        Returns [(image_id, {class_id: ndarray [n, 5]}), ...] — the reference's per-image result."""
        image_id, outputs, metas = outputs
        dets = [self.decode(o) for o in outputs]
        rows, counts = post.ctdet_merge(dets, metas, self.num_classes, self.down_ratio, self.test_max_per_image)
        return list(zip(image_id, post.results_by_class(rows, counts, self.num_classes)))

    def coco_rows(self, detections):
        """The aggregation half of test_epoch_end (centernet_detection.py:231-249): [(image_id, {class: [n, 5] x1 y1 x2 y2 score})]
        -> one float64 array [M, 7] = image id, x, y, w, h, score, COCO category id (`valid_ids`), the `loadRes` input format.
        The per-class boxes are NOT modified in place (the reference rewrites x2 / y2 of its inputs)."""
        import numpy as np
        data = []
        for image_id, detection in detections:
            for class_index, box in detection.items():
                box = np.asarray(box, dtype=np.float64)
                if box.shape[0] == 0:
                    continue
                out = np.empty((box.shape[0], 7), dtype=np.float64)
                out[:, 0] = image_id
                out[:, 1:3] = box[:, 0:2]
                out[:, 3] = box[:, 2] - box[:, 0]
                out[:, 4] = box[:, 3] - box[:, 1]
                out[:, 5] = box[:, 4]
                out[:, 6] = self.valid_ids[class_index - 1]
                data.append(out)
        return np.concatenate(data, axis=0) if data else np.zeros((0, 7), dtype=np.float64)

    def test_epoch_end(self, detections):
        """centernet_detection.py:227-265.  Without a COCO ground-truth handle the detections are returned unchanged (the
        reference's early return); with one, the rows go through `test_coco.loadRes` and pycocotools' COCOeval, and the six AP
        numbers are logged under the reference's names.  COCOeval itself is outside the hot path (out of
        scope) and is imported only here."""
        if not self.test_coco:
            return detections
        data = self.coco_rows(detections)
        coco_detections = self.test_coco.loadRes(data)
        from pycocotools.cocoeval import COCOeval      # optional dependency, as in the reference
        coco_eval = COCOeval(self.test_coco, coco_detections, "bbox")
        coco_eval.evaluate()
        coco_eval.accumulate()
        coco_eval.summarize()
        prefix = ("multi-scale_" if len(self.test_scales) > 1 else "") + ("flip_" if self.test_flip else "")
        for num, name in enumerate(["ap", "ap_50", "ap_75", "ap_S", "ap_M", "ap_L"]):
            self.log(f"test/{prefix}{name}", coco_eval.stats[num], sync_dist=True)
        return data
