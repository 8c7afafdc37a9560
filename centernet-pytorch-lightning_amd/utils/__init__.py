"""Helpers with the reference's module names (decode primitives, losses, test-step post-processing) on the HIP kernels."""
