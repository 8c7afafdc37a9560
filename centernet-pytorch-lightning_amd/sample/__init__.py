"""Ground-truth encoders (reference: CenterNet/sample/)."""
from .ctdet import CenterDetectionSample, encode_ctdet_batch  # noqa: F401
from .multi_pose import MultiPoseSample, encode_multi_pose_batch  # noqa: F401
