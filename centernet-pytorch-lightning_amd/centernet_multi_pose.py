"""`CenterNetMultiPose` (reference: CenterNet/centernet_multi_pose.py:29-321) — forward / loss / decode on the HIP path."""
import torch

from .centernet import CenterNet
from .decode.multi_pose import multi_pose_decode
from .models.heads import CenterHead
from .utils.decode import sigmoid_clamped
from .utils.losses import FocalLoss, RegL1Loss, RegWeightedL1Loss


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
        hm_loss = wh_loss = off_loss = kp_loss = hm_kp_loss = hm_offset_loss = 0
        num_stacks = len(outputs)
        for output in outputs:
            output["heatmap"] = sigmoid_clamped(output["heatmap"])
            output["heatmap_keypoints"] = sigmoid_clamped(output["heatmap_keypoints"])
            hm_loss = hm_loss + self.criterion(output["heatmap"], target["heatmap"])
            wh_loss = wh_loss + self.criterion_width_height(output["width_height"], target["regression_mask"],
                                                            target["indices"], target["width_height"])
            off_loss = off_loss + self.criterion_regression(output["regression"], target["regression_mask"],
                                                            target["indices"], target["regression"])
            kp_loss = kp_loss + self.criterion_keypoints(output["keypoints"], target["keypoints_mask"], target["indices"],
                                                         target["keypoints"])
            hm_kp_loss = hm_kp_loss + self.criterion_heatmap_keypoints(output["heatmap_keypoints"], target["heatmap_keypoints"])
            hm_offset_loss = hm_offset_loss + self.criterion_regression(
                output["heatmap_keypoints_offset"], target["heatmap_keypoints_mask"], target["heatmap_keypoints_indices"],
                target["heatmap_keypoints_offset"])
        h = self.hparams
        loss = (h.hm_weight * hm_loss + h.wh_weight * wh_loss + h.off_weight * off_loss + h.hp_weight * kp_loss
                + h.hm_hp_weight * hm_kp_loss + h.off_weight * hm_offset_loss) / num_stacks
        return loss, {"loss": loss, "hm_loss": hm_loss, "kp_loss": kp_loss, "hm_kp_loss": hm_kp_loss,
                      "hm_offset_loss": hm_offset_loss, "wh_loss": wh_loss, "off_loss": off_loss}

    @torch.no_grad()
    def decode(self, output, K=100):
        """The decode call of test_step_end (centernet_multi_pose.py:215-235)."""
        return multi_pose_decode(output["heatmap"].sigmoid_(), output["width_height"], output["keypoints"],
                                 reg=output["regression"], hm_hp=output["heatmap_keypoints"].sigmoid_(),
                                 hp_offset=output["heatmap_keypoints_offset"], K=K)
