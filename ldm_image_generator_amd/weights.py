"""Staleness keys for derived copies of parameters (packed conv filters, transposed / bf16 weight copies, the native plan).

A copy is valid while its source has the same storage (``data_ptr``), the same autograd version counter AND no optimizer has
stepped since: ``torch.optim.AdamW(fused=True)`` (and any optimizer writing through ``.data``) updates parameters WITHOUT bumping
``_version`` (measured on torch 2.10 / ROCm), so the version alone would leave every derived copy one step behind.  A process-wide
post-hook on ``Optimizer.step`` bumps ``GENERATION``; every cache key includes it.  In-place writes through ``.data`` by user code
remain invisible -- ``UNet.invalidate_caches()`` / ``bump()`` is the documented escape hatch."""
import torch

GENERATION = [0]


def bump(*_args, **_kwargs):
    GENERATION[0] += 1


def key(t):
    """Cache key of a derived copy of tensor ``t``."""
    return (t.data_ptr(), t._version, GENERATION[0])


from torch.optim.optimizer import register_optimizer_step_post_hook  # noqa: E402

register_optimizer_step_post_hook(bump)
