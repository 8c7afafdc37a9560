"""Backbones on the NHWC conv engine: msra_resnet, resnet_dcn, pose_dla_dcn, large_hourglass (state_dict compatible with the reference)."""
