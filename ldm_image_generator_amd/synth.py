"""Formula ("counter-based") synthetic weights and inputs.

There are no checkpoints for this model anywhere (the reference git-ignores
``*.pt``, /root/reference/.gitignore:134), so benchmarks, golden fixtures and
parity tests all run on synthetic weights.  To make those weights identical on
every box without shipping hundreds of megabytes, each tensor is a pure
function of (its state_dict key, its shape): a splitmix64 hash of the element
index, mapped to a uniform value in ``[-bound, bound)`` with
``bound = 1/sqrt(fan_in)`` -- the same scale as PyTorch's default Conv/Linear
initialisation that the reference's constructors apply (unet.py:77-85,
vae.py:57-58).  Only integer arithmetic and one exact int->float conversion are
used, so the values are bit-identical across machines, numpy versions and
thread counts.
"""
import zlib

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _MASK
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def uniform_pm1(key, numel, salt=0):
    """``numel`` float32 values in [-1, 1), a pure function of (key, salt)."""
    seed = np.uint64(zlib.crc32(key.encode()) & 0xFFFFFFFF) | (np.uint64(salt & 0xFFFFFFFF) << np.uint64(32))
    with np.errstate(over="ignore"):
        idx = np.arange(numel, dtype=np.uint64)
        h = _splitmix64(idx ^ _splitmix64(np.full(1, seed, dtype=np.uint64)))
    top = (h >> np.uint64(40)).astype(np.int64)          # 24 random bits: exact in float32
    return ((top - (1 << 23)).astype(np.float32)) / np.float32(1 << 23)


def formula_tensor(key, shape, bound=1.0, salt=0):
    numel = int(np.prod(shape)) if len(shape) else 1
    v = uniform_pm1(key, numel, salt) * np.float32(bound)
    return torch.from_numpy(v.reshape(shape).copy())


def gaussian_tensor(key, shape, salt=0):
    """Approximately N(0,1) values (sum of 4 uniforms, exact in fp32 ordering)."""
    numel = int(np.prod(shape))
    acc = np.zeros(numel, dtype=np.float32)
    for j in range(4):
        acc = acc + uniform_pm1(key, numel, salt * 4 + j)
    return torch.from_numpy((acc * np.float32(np.sqrt(3.0 / 4.0))).reshape(shape).copy())


def _fan_in(name, shape):
    if len(shape) == 4:
        if "upsamples" in name or name.startswith("decoder_last") or ".decoder_last" in name:
            # ConvTranspose2d weight [in, out, kh, kw]; torch's fan_in uses dim 1
            return shape[1] * shape[2] * shape[3]
        return shape[1] * shape[2] * shape[3]
    if len(shape) == 2:
        return shape[1]
    return None


def fill_state_dict(state_dict, salt=0, gain=1.0):
    """Return a new dict with every tensor of ``state_dict`` replaced by its formula value.

    Weights use bound = gain/sqrt(fan_in); biases use the fan_in of the weight
    that shares their prefix (falls back to their own length).
    """
    out = {}
    fans = {}
    for k, v in state_dict.items():
        f = _fan_in(k, tuple(v.shape))
        if f is not None:
            fans[k.rsplit(".", 1)[0]] = f
            if k.endswith("in_proj_weight"):
                fans[k[: -len("in_proj_weight")] + "in_proj"] = f
    for k, v in state_dict.items():
        shape = tuple(v.shape)
        prefix = k.rsplit(".", 1)[0]
        if k.endswith("in_proj_bias"):
            f = fans.get(k[: -len("in_proj_bias")] + "in_proj", shape[0])
        else:
            f = fans.get(prefix, shape[-1] if shape else 1)
        bound = gain / float(np.sqrt(max(f, 1)))
        out[k] = formula_tensor(k, shape, bound, salt).to(v.dtype)
    return out
