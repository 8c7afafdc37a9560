"""Counter-based deterministic generator (splitmix64), independent of torch's RNG.

The same numbers come out in the build container (where golden fixtures are
made from the imported reference) and on the GPU box (where inputs are
regenerated), so fixtures only need to hold *outputs*.
"""
import zlib

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _stream(seed, tag):
    if isinstance(tag, str):
        tag = zlib.crc32(tag.encode())
    return np.uint64((int(seed) * 0x100000001B3 + int(tag) * 0x9E3779B1) & 0xFFFFFFFFFFFFFFFF)


def bits(seed, tag, n):
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + _stream(seed, tag) * np.uint64(0x2545F4914F6CDD1D)
        return _splitmix64(ctr & _M64)


def uniform(seed, tag, shape, lo=0.0, hi=1.0):
    """float32 U[lo,hi) with 24 random mantissa bits (exactly representable)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (bits(seed, tag, n) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(seed, tag, shape, mean=0.0, std=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    m = (n + 1) // 2
    u1 = (bits(seed, (zlib.crc32(tag.encode()) if isinstance(tag, str) else tag) ^ 0x5BD1E995, m)
          >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    u2 = (bits(seed, tag, m) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    z = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])[:n]
    return (mean + std * z).astype(np.float32).reshape(shape)


def randint(seed, tag, shape, lo, hi):
    n = int(np.prod(shape)) if len(shape) else 1
    return (lo + (bits(seed, tag, n) % np.uint64(hi - lo)).astype(np.int64)).reshape(shape)


def t_uniform(seed, tag, shape, lo=0.0, hi=1.0):
    return torch.from_numpy(uniform(seed, tag, tuple(shape), lo, hi))


def t_normal(seed, tag, shape, mean=0.0, std=1.0):
    return torch.from_numpy(normal(seed, tag, tuple(shape), mean, std))


def fill_state_dict(module, seed=1234, bn_jitter=True, var_scale=1.0):
    """Deterministic weights for any module exposing the reference's state_dict names.

    conv / deconv weights ~ N(0, sqrt(2/fan_in)); biases small; BN gamma ~ U[0.8,1.2],
    beta ~ U[-0.1,0.1], running_mean ~ N(0,0.1), running_var ~ U[0.5,1.5].
    Heads follow the reference init (heads.py:45-50): ``heatmap*`` last bias -2.19.
    DCN ``conv_offset_mask`` gets small non-zero values so sampling is exercised.
    ``var_scale`` multiplies every running_var (eval-mode fixtures of very deep residual stacks: without it the
    ~50 un-normalised residual additions of Hourglass-104 double the variance each and the maps reach 1e7).
    """
    sd = module.state_dict()
    out = {}
    for name, t in sd.items():
        shape = tuple(t.shape)
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros_like(t)
            continue
        if not t.dtype.is_floating_point:
            out[name] = t.clone()
            continue
        leaf = name.split(".")[-1]
        if leaf == "running_mean":
            v = normal(seed, name, shape, 0.0, 0.1)
        elif leaf == "running_var":
            v = uniform(seed, name, shape, 0.5, 1.5) * np.float32(var_scale)
        elif t.dim() == 1 and leaf == "weight":      # BN gamma
            v = uniform(seed, name, shape, 0.8, 1.2) if bn_jitter else np.ones(shape, np.float32)
        elif t.dim() == 1:                           # bias / BN beta
            v = uniform(seed, name, shape, -0.1, 0.1)
            if ".fc.2.bias" in name and name.split(".")[-4].startswith("heatmap"):
                v = np.full(shape, -2.19, np.float32)
        elif "conv_offset_mask.weight" in name:
            v = normal(seed, name, shape, 0.0, 0.05 / np.sqrt(shape[1] * 9))
        elif t.dim() == 4 and (".up_" in name or name.split(".")[-2].startswith("up_")):
            # depthwise bilinear up-conv (pose_dla_dcn.py:424-432) + small jitter
            k = shape[2]
            f = int(np.ceil(k / 2)); c = (2 * f - 1 - f % 2) / (2.0 * f)
            g = np.array([1 - abs(i / f - c) for i in range(k)], np.float32)
            v = np.broadcast_to(np.outer(g, g), shape).copy()
            v += normal(seed, name, shape, 0.0, 0.01)
        elif t.dim() == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            if "deconv_layers" in name:              # ConvTranspose: [Ci,Co,k,k]; 4 taps hit each output
                fan_in = shape[0] * shape[2] * shape[3] / 4.0
            std = np.sqrt(2.0 / fan_in)
            if ".fc.2.weight" in name and name.split(".")[-4].startswith("heatmap"):
                std *= 0.002                         # keep heat-map logits near the -2.19 prior (no sigmoid saturation)
            v = normal(seed, name, shape, 0.0, std)
        else:
            v = normal(seed, name, shape, 0.0, 0.02)
        out[name] = torch.from_numpy(np.ascontiguousarray(v)).to(t.dtype)
    module.load_state_dict(out)
    return module
