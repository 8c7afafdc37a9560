"""Ground-truth encoders (reference: CenterNet/sample/).

`CenterDetectionSample` / `MultiPoseSample` keep the reference's per-image callable interface (numpy in, numpy out) on top of
the batch entry points `encode_ctdet_batch` / `encode_multi_pose_batch`, which build the dense targets of a whole minibatch on
the device with one launch of `cn_encode_ctdet` / `cn_encode_multi_pose` (csrc/encode.hip).
"""
from . import ctdet as _ctdet, multi_pose as _pose

CenterDetectionSample, encode_ctdet_batch = _ctdet.CenterDetectionSample, _ctdet.encode_ctdet_batch
MultiPoseSample, encode_multi_pose_batch = _pose.MultiPoseSample, _pose.encode_multi_pose_batch

__all__ = ["CenterDetectionSample", "MultiPoseSample", "encode_ctdet_batch", "encode_multi_pose_batch"]
