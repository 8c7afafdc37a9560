"""MI355X-native CenterNet hot path (backbone -> heads -> losses -> decode) behind the
plugin surface of tteepe/CenterNet-pytorch-lightning.  See DESIGN.md."""
__version__ = "0.1.0"


def __getattr__(name):  # lazy: importing the package must not require a GPU or the built .so
    if name in ("CenterNet", "CenterNetDetection", "CenterNetMultiPose"):
        import importlib
        mod = {"CenterNet": "centernet", "CenterNetDetection": "centernet_detection",
               "CenterNetMultiPose": "centernet_multi_pose"}[name]
        return getattr(importlib.import_module(f"{__name__}.{mod}"), name)
    raise AttributeError(name)
