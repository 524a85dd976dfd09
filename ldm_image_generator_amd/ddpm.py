"""DDPM training loss / DDIM sampler (reference: ddpm.py:11-93), MI355X-native host side.

Same constructor and method signatures as the reference.  Schedule tables stay
plain CPU tensors (not buffers -- ``state_dict`` is ``model.*`` only, like the
reference).  Per step the host computes the four DDIM scalars with the
reference's own fp32 expressions and the update runs as one fused elementwise
kernel that rounds exactly like the reference's separate torch ops.

Differences, all deliberate and documented in DESIGN.md:
  * ``model=None`` builds a fresh ``UNet()`` per instance (the reference evaluates
    ``UNet()`` once at import and shares it between instances, ddpm.py:16);
  * ``use_autocast`` selects reduced precision only after an explicit opt-in
    (``autocast.set_autocast_dtype(unet, torch.bfloat16)``): the default stays exact
    fp32, which is what the reference's CPU path computes (ddpm.py:75 is a no-op
    there); with the opt-in, ``use_autocast=True`` runs every UNet forward of the loop
    with bf16 GEMM operands (fp32 accumulate, fp32 residual stream, fp32 DDIM update);
  * ``sample(..., x_init=)`` optionally injects x_T (parity tests need a
    CPU-generated start; the reference draws it on the model's device).
"""
import random

import torch
import torch.nn as nn
from tqdm import tqdm

from . import ops
from .unet import UNet


class DDPM(nn.Module):
    def __init__(self, model=None, beta_min=1e-4, beta_max=0.02, num_timesteps=1000, loss_function=nn.L1Loss(),
                 lambda_max=20, lambda_min=-20):
        super().__init__()
        self.model = UNet() if model is None else model
        self.beta = torch.linspace(beta_min, beta_max, num_timesteps)
        self.alpha = 1 - self.beta
        self.num_timesteps = num_timesteps
        self.loss_function = loss_function
        self.lambda_max = lambda_max
        self.lambda_min = lambda_min
        # ddpm.py:28-37, same expressions (prefix products, then beta_tilde)
        self.alpha_bar = torch.Tensor([torch.prod(self.alpha[:t]) for t in range(1, num_timesteps + 1)])
        bt = [1]
        for t in range(1, num_timesteps):
            bt.append((1 - self.alpha_bar[t - 1]) / (1 - self.alpha_bar[t]) * self.beta[t])
        self.beta_tilde = torch.Tensor(bt)

    def calculate_loss(self, x, condition=None):
        """ddpm.py:39-48.  UNet forward AND backward run on the HIP path (train.py)."""
        t = torch.randint(low=1, high=self.num_timesteps, size=(x.shape[0],))
        alpha_bar_t = torch.index_select(self.alpha_bar, 0, t)
        e = torch.randn(*x.shape, device=x.device)
        # t and the two factors are drawn / computed on the host like the reference's; they go to the GPU through pinned memory without
        # blocking: a plain .to(device) waits for the stream to drain, i.e. for the whole previous training step, and the GPU then
        # idles while the host prepares this one (5 ms per step at the default widths)
        def up(v):
            return v.pin_memory().to(x.device, non_blocking=True) if x.is_cuda else v.to(x.device)
        sa = up(torch.sqrt(alpha_bar_t))
        sb = up(torch.sqrt(1 - alpha_bar_t))
        xt = torch.empty_like(x, dtype=torch.float32)
        ops.qsample(x.contiguous().float(), e, sa, sb, xt)
        t = up(t)
        e_theta = self.model(x=xt, time=t, condition=condition)
        if type(self.loss_function) is nn.L1Loss and self.loss_function.reduction == "mean":
            from .train import L1LossFunction          # the default loss, fused fwd/bwd kernels
            return L1LossFunction.apply(e_theta, e)
        return self.loss_function(e_theta, e)

    @torch.no_grad()
    def sample(self, x_shape=(1, 3, 64, 64), condition=None, seed=None, num_steps=20, use_autocast=True, schedule='linear',
               eta=0, x_init=None, progress=True, shard=None):
        device = next(self.model.parameters()).device
        if seed is not None:
            random.seed(seed)
            torch.manual_seed(seed)
            torch.cuda.manual_seed(seed)
        x = torch.randn(*x_shape, device=device)                 # drawn even when x_init is given (RNG parity)
        if x_init is not None:
            x = x_init.to(device=device, dtype=torch.float32).clone()
        if schedule == 'linear':
            steps = [int(s) for s in torch.linspace(0, self.num_timesteps - 1, num_steps).int()]
        elif type(schedule) == list:
            steps = schedule
        else:
            raise TypeError(f"schedule \"{schedule}\" is not implemented.")   # the reference raises a str -> TypeError
        steps_next = [0] + steps[:-1]
        alpha = torch.cumprod((1 - self.beta), dim=0)
        bar = tqdm(total=len(steps), disable=not progress)
        hint_ok = hasattr(self.model, "_uniform_time")
        if hasattr(self.model, "_films_key"):
            self.model._films_key = None
        # ddpm.py:75 `with torch.cuda.amp.autocast(enabled=use_autocast)`: honoured when the model has opted into a 16-bit type
        amp = bool(use_autocast) and getattr(self.model, "autocast_dtype", None) is not None
        if amp:
            self.model._autocast_now = True
        t_dev = torch.tensor([int(s) for s in steps], dtype=torch.int64, device=device)   # one upload for the whole loop
        try:
            for i, (t, t_next) in enumerate(zip(reversed(steps), reversed(steps_next))):
                t_tensor = torch.full((x_shape[0],), t, device=device)
                if hint_ok:                                      # every sample shares t: FiLM computed once
                    k = len(steps) - 1 - i
                    self.model._uniform_time = (t, t_dev[k:k + 1], t_dev, k)  # (value, its device tensor, the whole schedule, index)
                e_theta = self.model(x=x, time=t_tensor, condition=None)
                e = torch.randn(*x_shape, device=device)
                if shard is not None and eta != 0:
                    # data-parallel sampling with eta > 0: the per-step noise of the GLOBAL batch comes from one CPU generator and
                    # every rank takes its slice (as for x_T), so sharded == unsharded and ranks never share noise
                    gb, lo, hi = shard
                    gen = torch.Generator().manual_seed((0 if seed is None else int(seed)) * 1000003 + i + 1)
                    e = torch.randn(gb, *x_shape[1:], generator=gen)[lo:hi].to(device)
                sigma = eta * torch.sqrt((1 - alpha[t_next]) / (1 - alpha[t])) * torch.sqrt(1 - alpha[t] / alpha[t_next])
                s1 = torch.sqrt(1 - alpha[t])
                s2 = torch.sqrt(alpha[t])
                s3 = torch.sqrt(alpha[t_next])
                s4 = torch.sqrt(1 - alpha[t_next] - sigma ** 2)
                bar.set_description(f"t: {t}, sigma: {sigma}")
                ops.ddim_update(x, e_theta, e, float(s1), float(s2), float(s3), float(s4), float(sigma), t == 0)
                bar.update(1)
        finally:
            if amp:
                self.model._autocast_now = False
            if hint_ok:
                self.model._uniform_time = None
                if hasattr(self.model, "_films_key"):
                    self.model._films_key = None              # the tables in the workspace belong to this loop only
            bar.close()
        return x
