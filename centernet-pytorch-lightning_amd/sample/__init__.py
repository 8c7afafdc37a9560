"""Ground-truth encoders (reference: CenterNet/sample/)."""
from .ctdet import CenterDetectionSample, encode_ctdet_batch  # noqa: F401
