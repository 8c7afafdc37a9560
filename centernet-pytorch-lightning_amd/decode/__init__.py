"""Fused decoders: ctdet_decode and multi_pose_decode (one launch chain per batch, csrc/decode.hip)."""
