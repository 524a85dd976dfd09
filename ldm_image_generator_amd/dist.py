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
# data-parallel training (BASELINE cfg 5): replicas + the gradient all-reduce of every step, bucketed per UNet level and overlapped
# with the backward (GradSync); allreduce_gradients is the flat, blocking form kept as the reference the buckets are tested against
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
    nccl = flat.is_cuda and dist.get_backend() == "nccl"
    dist.all_reduce(flat, op=dist.ReduceOp.AVG if nccl else dist.ReduceOp.SUM)
    if not nccl:
        flat /= world                                                # gloo (CPU tests, one-GPU rehearsals) has no AVG
    off = 0
    for p in used:
        n = p.numel()
        p.grad = flat[off:off + n].view_as(p)
        off += n
    return len(used)


class GradSync:
    """Bucketed gradient averaging overlapped with the backward ("one gradient all-reduce ... per step", here cut into
    one bucket per UNet level so that the collective of a finished level runs while the backward of the next one computes).

    ``UNetFunction.backward`` calls ``push`` each time the gradients of a level are final (decoder level 0 first, the stem last);
    a bucket is ONE flat fp32 buffer of that level's used gradients, reduced on a side stream (RCCL) behind an event of the compute
    stream; ``finish`` (end of the backward) waits for the buckets and hands the averaged views back.  No flat copy of the whole
    1.2 GB gradient exists.  Every rank seeds Python's ``random`` identically per step, so every rank builds the same buckets.

    ``wire_dtype=torch.bfloat16`` halves the bytes on the wire without summing in bf16: each rank sends shard j of its bucket,
    rounded to bf16, to rank j (all-to-all), rank j adds the ``world`` shards in fp32 in rank order, rounds the average to bf16 once
    and the shards are all-gathered.  The default (None) is the plain fp32 all-reduce.
    """

    def __init__(self, world, wire_dtype=None):
        self.world = world
        self.wire_dtype = wire_dtype
        self.pending = []
        self.stream = None
        self.exposed_ms = 0.0
        self.total_bytes = 0
        self._t = None

    def _side_stream(self, device):
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=device)
        return self.stream

    def _reduce(self, flat):
        """-> (work handle or None, result tensor)"""
        if self.wire_dtype is None:
            if flat.is_cuda and dist.get_backend() == "nccl":
                return dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=True), flat
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)        # gloo has no AVG
            return work, flat
        w = self.world
        n = flat.numel()
        pad = (-n) % (8 * w)
        send = torch.zeros(n + pad, dtype=self.wire_dtype, device=flat.device)
        send[:n] = flat                                                # one rounding to the wire type
        stage = flat.is_cuda and dist.get_backend() != "nccl"           # gloo rehearsals on one GPU: these two collectives take host tensors
        if stage:
            send = send.cpu()
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send)
        parts = recv.view(w, -1)
        acc = parts[0].float()
        for r in range(1, w):                                           # fp32 accumulation, rank order
            acc += parts[r].float()
        acc /= w
        mine = acc.to(self.wire_dtype)
        gathered = torch.empty_like(send)
        dist.all_gather_into_tensor(gathered, mine)
        flat.copy_(gathered[:n].to(flat.device))
        return None, flat

    def push(self, items, sink):
        """items: [(parameter, gradient)] of one finished level; sink: the dict the averaged gradients are written back to."""
        if not items:
            return
        flat = torch.cat([g.reshape(-1) for _, g in items])
        self.total_bytes += flat.numel() * (2 if self.wire_dtype is not None else 4)
        if flat.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            side = self._side_stream(flat.device)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                work, res = self._reduce(flat)
            flat.record_stream(side)
        else:
            work, res = self._reduce(flat)
        self.pending.append((items, res, work, sink))

    def finish(self):
        """Wait for every bucket and write the averaged gradients back; records how long the step waited here."""
        import time
        cuda = bool(self.pending) and self.pending[0][1].is_cuda
        if cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        t0 = time.perf_counter()
        for items, flat, work, sink in self.pending:
            if work is not None:
                work.wait()
            if flat.is_cuda:
                torch.cuda.current_stream().wait_stream(self.stream)
            if self.wire_dtype is None and not (flat.is_cuda and dist.get_backend() == "nccl"):
                flat /= self.world
            off = 0
            for p, g in items:
                n = g.numel()
                sink[p] = flat[off:off + n].view(g.shape)
                off += n
        if cuda:
            e1.record()
            self._t = (e0, e1)
        else:
            self.exposed_ms = (time.perf_counter() - t0) * 1e3
        self.pending = []

    def exposed(self):
        """Milliseconds the compute stream waited for the collectives at the end of the last backward (synchronises)."""
        if self._t is not None:
            self._t[1].synchronize()
            self.exposed_ms = self._t[0].elapsed_time(self._t[1])
            self._t = None
        return self.exposed_ms


WIRE_DTYPE = None        # torch.bfloat16: bf16 on the wire, fp32 accumulation (GradSync); None: fp32 all-reduce
BUCKETED = True          # False: the flat, blocking all-reduce after the backward (the form the buckets are tested against)


def train_step(ddpm, optimizer, x_local, step_seed, world, stats=None):
    """One optimisation step of train_ldm.py:76-86 on this rank's shard of the global batch.  ``stats`` (a dict) receives
    ``allreduce_ms_exposed`` and ``allreduce_bytes`` for N > 1."""
    import random
    random.seed(step_seed)                                           # identical expert / depth decisions on all ranks
    # ... but DIFFERENT timesteps and noise per rank (ddpm.py:40,44 draw them from torch's generators): a caller that seeds torch
    # identically on every rank would otherwise train all shards on the same (t, e)
    rank = dist.get_rank() if (world > 1 and dist.is_initialized()) else 0
    torch.manual_seed(step_seed * world + rank)
    optimizer.zero_grad()
    model = getattr(ddpm, "model", None)
    sync = GradSync(world, WIRE_DTYPE) if (world > 1 and BUCKETED and hasattr(model, "_grad_sync")) else None
    if sync is not None:
        model._grad_sync = sync                                      # UNetFunction.backward reduces level by level, overlapped
    try:
        loss = ddpm.calculate_loss(x_local)
        loss.backward()
    finally:
        if sync is not None:
            model._grad_sync = None
    if sync is None:
        allreduce_gradients(list(ddpm.parameters()), world)
    elif stats is not None:
        stats["allreduce_ms_exposed"] = sync.exposed()
        stats["allreduce_bytes"] = sync.total_bytes
    optimizer.step()
    return loss
