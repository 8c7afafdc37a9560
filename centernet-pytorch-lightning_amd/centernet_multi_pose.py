"""`CenterNetMultiPose` (reference: CenterNet/centernet_multi_pose.py:29-321) — forward / loss / decode / test step on the HIP path."""
import torch

from .centernet import CenterNet
from .decode.multi_pose import multi_pose_decode
from .models.heads import CenterHead
from .utils.decode import sigmoid_clamped
from .utils.losses import FocalLoss, RegL1Loss, RegWeightedL1Loss
from .utils import post


class CenterNetMultiPose(CenterNet):
    mean = [0.408, 0.447, 0.470]
    std = [0.289, 0.274, 0.278]
    flip_idx = [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15]

    def __init__(self, arch, learning_rate=1e-4, learning_rate_milestones=None, hm_weight=1, wh_weight=0.1, off_weight=1,
                 hp_weight=1, hm_hp_weight=1, test_coco=None, test_coco_ids=None, test_scales=None, test_flip=True,
                 compute_dtype=torch.bfloat16):
        super().__init__(arch, compute_dtype=compute_dtype)
        heads = {"heatmap": 1, "width_height": 2, "regression": 2, "heatmap_keypoints": 17, "keypoints": 34,
                 "heatmap_keypoints_offset": 2}
        self.heads = torch.nn.ModuleList([CenterHead(heads, self.backbone.out_channels, self.head_conv)
                                          for _ in range(self.num_stacks)])
        self.learning_rate_milestones = learning_rate_milestones if learning_rate_milestones is not None else []
        self.test_coco, self.test_coco_ids = test_coco, test_coco_ids
        self.test_max_per_image = 20
        self.test_scales = [1] if test_scales is None else test_scales
        self.test_flip = test_flip
        self.criterion = FocalLoss()
        self.criterion_heatmap_keypoints = FocalLoss()
        self.criterion_keypoints = RegWeightedL1Loss()
        self.criterion_regression = RegL1Loss()
        self.criterion_width_height = RegL1Loss()
        self.save_hyperparameters()

    def forward(self, x):
        return [head(out) for head, out in zip(self.heads, self.backbone(x))]

    def loss(self, outputs, target):
        hm, wh, off, kp, hm_kp, hm_off = [], [], [], [], [], []
        num_stacks = len(outputs)
        for output in outputs:
            output["heatmap"], t = self._sigmoid_focal(self.criterion, output["heatmap"], target["heatmap"])
            output["heatmap_keypoints"], t_kp = self._sigmoid_focal(self.criterion_heatmap_keypoints, output["heatmap_keypoints"],
                                                                    target["heatmap_keypoints"])
            hm.append(t)
            wh.append(self.criterion_width_height(output["width_height"], target["regression_mask"], target["indices"],
                                                  target["width_height"]))
            off.append(self.criterion_regression(output["regression"], target["regression_mask"], target["indices"],
                                                 target["regression"]))
            kp.append(self.criterion_keypoints(output["keypoints"], target["keypoints_mask"], target["indices"], target["keypoints"]))
            hm_kp.append(t_kp)
            hm_off.append(self.criterion_regression(output["heatmap_keypoints_offset"], target["heatmap_keypoints_mask"],
                                                    target["heatmap_keypoints_indices"], target["heatmap_keypoints_offset"]))
        h = self.hparams
        groups = [hm, wh, off, kp, hm_kp, hm_off]
        hm_loss, wh_loss, off_loss, kp_loss, hm_kp_loss, hm_offset_loss = (
            ts[0] if num_stacks == 1 else self._sum_terms(ts) for ts in groups)
        # centernet_multi_pose.py:126-140: the weighted total / num_stacks, one launch (ops.weighted_sum)
        loss = self._weighted_total(groups, [h.hm_weight, h.wh_weight, h.off_weight, h.hp_weight, h.hm_hp_weight, h.off_weight], num_stacks)
        return loss, {"loss": loss, "hm_loss": hm_loss, "kp_loss": kp_loss, "hm_kp_loss": hm_kp_loss,
                      "hm_offset_loss": hm_offset_loss, "wh_loss": wh_loss, "off_loss": off_loss}

    @torch.no_grad()
    def decode(self, output, K=100):
        """The decode call of test_step_end (centernet_multi_pose.py:215-235)."""
        return multi_pose_decode(output["heatmap"].sigmoid_(), output["width_height"], output["keypoints"],
                                 reg=output["regression"], hm_hp=output["heatmap_keypoints"].sigmoid_(),
                                 hp_offset=output["heatmap_keypoints_offset"], K=K)

    def _flip_tables(self, device):
        """perm / sign vectors of the two pose-aware merges (centernet_multi_pose.py:203-210), cached on the device."""
        key = str(device)
        if getattr(self, "_flip_cache", None) is None or self._flip_cache[0] != key:
            idx = torch.tensor(self.flip_idx, dtype=torch.int32)
            kp_perm = torch.stack([2 * idx, 2 * idx + 1], 1).flatten()                 # channel 2j+xy <- 2*flip_idx[j]+xy
            kp_sign = torch.tensor([-1.0, 1.0]).repeat(len(self.flip_idx))             # x components are negated
            self._flip_cache = (key, kp_perm.to(device), kp_sign.to(device), idx.to(device),
                                torch.ones(len(self.flip_idx), device=device))
        return self._flip_cache[1:]

    @torch.no_grad()
    def test_step(self, batch, batch_idx):
        """centernet_multi_pose.py:157-213 for a BATCH of images in [0, 1]: per scale pad / normalise / mirror in one launch,
        forward, then the mirrored maps folded back — box maps averaged, keypoint maps with the left/right joint swap."""
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
                kp_perm, kp_sign, hm_perm, hm_sign = self._flip_tables(x.device)
                out = {"heatmap": post.flip_merge(out["heatmap"]), "width_height": post.flip_merge(out["width_height"]),
                       "regression": out["regression"][:B].contiguous(),
                       "keypoints": post.flip_merge_perm(out["keypoints"], kp_perm, kp_sign),
                       "heatmap_keypoints": post.flip_merge_perm(out["heatmap_keypoints"], hm_perm, hm_sign),
                       "heatmap_keypoints_offset": out["heatmap_keypoints_offset"][:B].contiguous()}
            outputs.append(out)
            meta.append({"scale": [nw / width, nh / height], "padding": [pad_x, pad_y]})
        return image_id, outputs, meta

    @torch.no_grad()
    def test_step_end(self, outputs):
        """centernet_multi_pose.py:213-264 for the batch: decode per scale, ONE launch for rescaling / soft_nms_39 / the
        max-per-image cut, one host copy.  Returns [(image_id, rows as nested lists [n][57]), ...]."""
        image_id, outputs, metas = outputs
        dets = [self.decode(o) for o in outputs]
        rows, counts = post.pose_merge(dets, metas, self.down_ratio, self.test_max_per_image)
        rows, counts = rows.cpu().numpy(), counts.cpu().numpy()
        return [(i, rows[b, :counts[b]].tolist()) for b, i in enumerate(image_id)]

    @staticmethod
    def coco_annotations(results):
        """The aggregation half of test_epoch_end (centernet_multi_pose.py:266-296): [(image_id, rows [n][>= 39])] -> the list of COCO
        keypoint-annotation dicts `loadRes` takes: bbox as x, y, w, h, 17 keypoints as (x, y, 1) triples, category 1."""
        import numpy as np
        data = []
        for image_id, detections in results:
            for detection in detections:
                d = np.asarray(detection, dtype=np.float64)
                bbox = [float(d[0]), float(d[1]), float(d[2] - d[0]), float(d[3] - d[1])]
                kps = np.concatenate([d[5:39].astype(np.float32).reshape(-1, 2), np.ones((17, 1), dtype=np.float32)], axis=1)
                data.append({"image_id": int(image_id), "category_id": 1, "bbox": bbox, "score": float(d[4]),
                             "keypoints": kps.reshape(51).tolist()})
        return data

    def test_epoch_end(self, results):
        """centernet_multi_pose.py:266-318: nothing without a COCO handle (the reference's early return); with one, keypoint and box
        AP through pycocotools' COCOeval (imported only here: out of the hot path's scope), logged under the reference's names."""
        if not self.test_coco:
            return None
        data = self.coco_annotations(results)
        coco_detections = self.test_coco.loadRes(data)
        from pycocotools.cocoeval import COCOeval
        evals = {}
        for kind, tag in (("keypoints", "kp"), ("bbox", "bbox")):
            ev = COCOeval(self.test_coco, coco_detections, kind)
            ev.evaluate(); ev.accumulate(); ev.summarize()
            evals[tag] = ev
        prefix = ("multi-scale_" if len(self.test_scales) > 1 else "") + ("flip_" if self.test_flip else "")
        for tag in ("kp", "bbox"):
            for num, name in enumerate(["ap", "ap_50", "ap_75", "ap_S", "ap_M", "ap_L"]):
                self.log(f"test/{tag}_{prefix}{name}", evals[tag].stats[num], sync_dist=True)
        return data
