"""Opt-in reduced precision for sampling and decoding.

On a GPU the reference runs ``DDPM.sample`` under 16-bit autocast by default (ddpm.py:52 ``use_autocast=True``, ddpm.py:75,
sample_ldm.py:17,72).  The HIP path is exact fp32 unless a module opts in here; fp16 overflows on these weights,
so the 16-bit type is bf16:

    autocast.set_autocast_dtype(unet, torch.bfloat16)   # DDPM.sample(use_autocast=True) then runs bf16 GEMM operands
    autocast.set_compute_dtype(decoder, torch.bfloat16) # Decoder.forward keeps its activations as bf16 rows

The Decoder switch is an extension: the reference decodes outside its autocast region (sample_ldm.py:73-74).  ``None`` restores
exact fp32.  Accuracy of the bf16 mode against the reference's fp32 goldens is stated in tests/test_gpu_autocast.py.
"""
import torch

_ALLOWED = (None, torch.bfloat16)


def set_autocast_dtype(unet, dtype):
    if dtype not in _ALLOWED:
        raise ValueError("autocast dtype must be None or torch.bfloat16 (fp16 overflows on this model), got %r" % (dtype,))
    unet.autocast_dtype = dtype
    return unet


def set_compute_dtype(decoder, dtype):
    if dtype not in _ALLOWED:
        raise ValueError("compute dtype must be None or torch.bfloat16, got %r" % (dtype,))
    decoder.compute_dtype = dtype
    return decoder
