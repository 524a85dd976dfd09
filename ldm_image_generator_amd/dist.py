"""Data-parallel sampling: shard the sample batch, one all-gather at the end.

Samples never interact anywhere on the path (per-pixel norm, per-window attention,
no batch statistics -- SURVEY.md 8e), so rank r simply owns the contiguous slice
[r*B/N, (r+1)*B/N) of the global batch; weights are replicated; every rank seeds
Python's ``random`` identically so expert/skip decisions equal the unsharded run.
There is NO collective inside the 50-step loop or the decode; the only exchange is
one all-gather of the finished images (RCCL over xGMI when the backend is "nccl").
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:      # LDM_DIST_BACKEND=gloo: rehearse the multi-rank control flow where RCCL cannot run (one GPU, CPU)
            backend = os.environ.get("LDM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(global_batch, rank, world):
    """Contiguous, balanced slices (the first ``global_batch % world`` ranks get one extra)."""
    base, extra = divmod(global_batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def global_noise(global_batch, latent_shape, seed):
    """x_T for the WHOLE job from one CPU generator, so sharded == unsharded sample-for-sample."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(global_batch, *latent_shape, generator=g)


def gather_images(local_images, global_batch, rank, world):
    """One all-gather of the per-rank results into [global_batch, ...] on every rank."""
    if world == 1:
        return local_images
    base, extra = divmod(global_batch, world)
    if extra == 0:
        out = torch.empty((global_batch,) + tuple(local_images.shape[1:]), dtype=local_images.dtype, device=local_images.device)
        dist.all_gather_into_tensor(out, local_images.contiguous())
        return out
    # ragged: pad every shard to the largest, gather, then drop the padding
    big = base + 1
    pad = torch.zeros((big,) + tuple(local_images.shape[1:]), dtype=local_images.dtype, device=local_images.device)
    pad[: local_images.shape[0]] = local_images
    out = torch.empty((world * big,) + tuple(local_images.shape[1:]), dtype=local_images.dtype, device=local_images.device)
    dist.all_gather_into_tensor(out, pad)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(global_batch, r, world)
        parts.append(out[r * big: r * big + (hi - lo)])
    return torch.cat(parts, dim=0)


def sample_images_sharded(sample_fn, decode_fn, global_batch, latent_shape, seed, rank, world, device, gather=True,
                          as_uint8=False):
    """Run ``decode_fn(sample_fn(x_T_slice))`` on this rank's slice and all-gather the images.

    sample_fn(x_T [b, ...] on device) -> latents; decode_fn(latents) -> images.  Every
    rank must call with the same (global_batch, latent_shape, seed).  ``as_uint8`` applies the reference's
    post-process (sample_ldm.py:75-77: clamp, scale, truncate, HWC) on the device BEFORE the gather, so the one
    collective moves [B, H, W, 3] uint8 -- a quarter of the fp32 message.
    """
    lo, hi = shard_bounds(global_batch, rank, world)
    x_t = global_noise(global_batch, latent_shape, seed)[lo:hi].to(device)
    images = decode_fn(sample_fn(x_t))
    if as_uint8:
        from .vae import to_uint8_images
        images = to_uint8_images(images)
    return gather_images(images, global_batch, rank, world) if gather else images


# ------------------------------------------------------------------------------------------------------
# data-parallel training (BASELINE cfg 5): replicas + ONE gradient all-reduce per step
# ------------------------------------------------------------------------------------------------------
def allreduce_gradients(params, world):
    """Average the gradients of the parameters that took part in this step with a single flat all-reduce.

    Every rank seeds Python's ``random`` identically before ``calculate_loss`` (the reference draws ONE set of
    stochastic-depth / expert decisions per step for its whole batch), so the set of parameters with a
    gradient is the same on every rank and no "used" flags have to travel.  Unused parameters keep
    ``grad is None`` and are skipped by AdamW exactly as in the single-process reference.
    """
    used = [p for p in params if p.grad is not None]
    if world == 1 or not used:
        return len(used)
    # ONE bucket (<= 1.5 GB of fp32 gradients): xGMI is point-to-point, so one large ring all-reduce beats many small ones; the
    # copy into / out of the flat buffer is 2 x 1.5 GB of HBM traffic (< 1 ms).  Not overlapped with the backward.
    flat = torch.cat([p.grad.reshape(-1) for p in used])
    dist.all_reduce(flat, op=dist.ReduceOp.AVG if flat.is_cuda else dist.ReduceOp.SUM)
    if not flat.is_cuda:
        flat /= world                                                # gloo (CPU tests) has no AVG
    off = 0
    for p in used:
        n = p.numel()
        p.grad = flat[off:off + n].view_as(p)
        off += n
    return len(used)


def train_step(ddpm, optimizer, x_local, step_seed, world):
    """One optimisation step of train_ldm.py:76-86 on this rank's shard of the global batch."""
    import random
    random.seed(step_seed)                                           # identical expert / depth decisions on all ranks
    # ... but DIFFERENT timesteps and noise per rank (ddpm.py:40,44 draw them from torch's generators): a caller that seeds torch
    # identically on every rank would otherwise train all shards on the same (t, e)
    rank = dist.get_rank() if (world > 1 and dist.is_initialized()) else 0
    torch.manual_seed(step_seed * world + rank)
    optimizer.zero_grad()
    loss = ddpm.calculate_loss(x_local)
    loss.backward()
    allreduce_gradients(list(ddpm.parameters()), world)
    optimizer.step()
    return loss
