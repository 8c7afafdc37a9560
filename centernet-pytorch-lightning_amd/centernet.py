"""`CenterNet` base task module (reference: CenterNet/centernet.py:9-119).

Subclasses `pytorch_lightning.LightningModule` when Lightning is importable; otherwise a minimal stand-in with
the members the reference uses (`log`, `hparams`, `save_hyperparameters`) so the same class also runs under
this package's own trainer (engine.Trainer) — Lightning is not in the offline image.
"""
import inspect
from argparse import ArgumentParser, Namespace

import torch

from .models import create_model

try:  # pragma: no cover - Lightning is absent in the build image
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    pl = None

    class _Base(torch.nn.Module):
        """The slice of LightningModule the CenterNet task modules touch."""

        def __init__(self):
            super().__init__()
            self.hparams = Namespace()
            self.logged = {}

        def log(self, name, value, **kwargs):
            self.logged[name] = value

        def save_hyperparameters(self):
            frame = inspect.currentframe().f_back
            args = inspect.getargvalues(frame)
            for k in args.args:
                if k != "self":
                    setattr(self.hparams, k, args.locals[k])


class CenterNet(_Base):
    def __init__(self, arch, compute_dtype=torch.bfloat16):
        super().__init__()
        self.arch = arch
        # centernet.py:15-17
        self.head_conv = 256 if "dla" in arch or "hourglass" in arch else 64
        self.num_stacks = 2 if "hourglass" in arch else 1
        self.padding = 127 if "hourglass" in arch else 31
        self.backbone = create_model(arch, compute_dtype=compute_dtype, nchw_out=False)     # NHWC handle straight into the heads
        self.down_ratio = 4

    @staticmethod
    def _sigmoid_focal(criterion, x, target):
        """`y = sigmoid_clamped(x); loss = criterion(y, target)` (centernet_detection.py:103-106); one autograd node with a
        single-pass backward when the criterion is this package's FocalLoss."""
        if hasattr(criterion, "on_logits"):
            return criterion.on_logits(x, target)
        from .utils.decode import sigmoid_clamped
        y = sigmoid_clamped(x)
        return y, criterion(y, target)

    @staticmethod
    def _sum_terms(ts):
        from . import ops
        return ops.weighted_sum(ts, [1.0] * len(ts))

    @staticmethod
    def _weighted_total(groups, weights, num_stacks):
        """sum_g weights[g] * sum(groups[g]) / num_stacks as one device launch (<= 8 terms), else in chunks"""
        from . import ops
        ts = [t for g in groups for t in g]
        ws = [w / num_stacks for g, w in zip(groups, weights) for _ in g]
        while len(ts) > 8:
            ts, ws = [ops.weighted_sum(ts[:8], ws[:8])] + ts[8:], [1.0] + ws[8:]
        return ops.weighted_sum(ts, ws)

    @property
    def compute_dtype(self):
        return self.backbone.compute_dtype

    @compute_dtype.setter
    def compute_dtype(self, dt):
        self.backbone.compute_dtype = dt

    def load_pretrained_weights(self, model_weight_path, strict=True):
        """centernet.py:23-62: remap an original-CenterNet checkpoint (hm/wh/reg/... heads) onto backbone + heads."""
        mapping = {"hm": "heatmap", "wh": "width_height", "reg": "regression", "hm_hp": "heatmap_keypoints",
                   "hp_offset": "heatmap_keypoints_offset", "hps": "keypoints"}
        print(f"Loading weights from: {model_weight_path}")
        sd = torch.load(model_weight_path, map_location="cpu")["state_dict"]
        strip = lambda k: k.replace("module.", "")
        self.backbone.load_state_dict({strip(k): v for k, v in sd.items() if k.split(".")[1] not in mapping}, strict=strict)
        heads = {}
        for k, v in sd.items():
            parts = k.split(".")
            if parts[1] not in mapping:
                continue
            name = ".".join([mapping[strip(k).split(".")[0]], "fc"] + parts[2:]).replace("conv.", "")
            heads[("0." if self.num_stacks == 1 else "") + name] = v
        if self.arch == "hourglass":
            # centernet.py:55-61: the original hourglass heads are `<head>.<stack>.<layer>`; here `<stack>.<head>.fc.<layer>`,
            # and the second conv sits in slot 2 of `fc` (slot 1 is the ReLU)
            def restack(k):
                p = k.split(".")
                return ".".join(p[2:3] + p[:2] + p[3:]).replace("fc.1", "fc.2")
            heads = {restack(k): v for k, v in heads.items()}
        self.heads.load_state_dict(heads, strict=strict)

    def forward(self, x):
        return self.backbone.forward(x)

    def loss(self, outputs, target):
        return 0, {}

    def _loss_and_log(self, batch, stage):
        """forward + loss + the reference's log keys (`<stage>_loss`, `<stage>/<term>`; validation logs per epoch, synced)."""
        img, target = batch
        loss, terms = self.loss(self(img), target)
        per_epoch = {"on_epoch": True, "sync_dist": True} if stage == "val" else {}
        self.log(f"{stage}_loss", loss, **{"on_epoch": True, **per_epoch})
        for term, value in terms.items():
            self.log(f"{stage}/{term}", value, **per_epoch)
        return loss, terms

    def training_step(self, batch, batch_idx):
        """centernet.py:70-80."""
        return self._loss_and_log(batch, "train")[0]

    def validation_step(self, batch, batch_idx):
        """centernet.py:82-92."""
        loss, terms = self._loss_and_log(batch, "val")
        return {"loss": loss, "loss_stats": terms}

    def configure_optimizers(self):
        """centernet.py:94-105: Adam(lr) + MultiStepLR(milestones) stepped per epoch.  The optimizer is this package's
        fused flat-buffer Adam (engine.FlatAdam -> cn_adam_step) exposing the torch.optim interface."""
        from .engine import FlatAdam
        optimizer = FlatAdam(self.parameters(), lr=self.hparams.learning_rate)
        lr_scheduler = {"scheduler": torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=self.learning_rate_milestones),
                        "name": "learning_rate", "interval": "epoch", "frequency": 1}
        return [optimizer], [lr_scheduler]

    @staticmethod
    def add_model_specific_args(parent_parser):
        parser = ArgumentParser(parents=[parent_parser], add_help=False)
        parser.add_argument("--arch", default="dla_34", help="backbone architecture: res_18 | res_101 | resdcn_18 | resdcn_101 | dla_34 | hourglass")
        parser.add_argument("--learning_rate", type=float, default=25e-5)
        parser.add_argument("--learning_rate_milestones", default="90, 120")
        return parser
