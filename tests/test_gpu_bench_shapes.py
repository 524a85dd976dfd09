"""Size-independent properties of the HIP path AT THE SHAPES bench.py TIMES (BASELINE.json configs[2] and configs[4]):

  * sampling  [256, 8, 32, 32], eval mode, 50 DDIM steps + VAE decode: rows of the batch re-run alone equal the batch's rows
    (samples never interact) -- the M = 262 144-row launches take ring / stream / band paths no small test reaches;
  * training  [128, 8, 64, 64] (cfg 5 per-GPU shape), fp32 and bf16 operands: the step equals the mean of its two 64-sample halves
    under one ``random.seed`` (mean-reduced L1 loss, no batch statistics) -- covers the ring auto-dispatch, the TN split counts at
    M = 524 288 and the partial-plane reductions of the timed step.

Stated tolerances: sampling rows rel-L2 <= 5e-6 (fp32 re-association between tile paths; measured 7.3e-7); training step <= 2e-4
per gradient tensor in both precisions (measured 1.3e-6 fp32, 4.9e-7 bf16: every per-sample value is rounded to bf16 identically in the
whole batch and in its halves -- only the fp32 sums over samples re-associate).
"""
import random

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def formula(module):
    from ldm_image_generator_amd import synth
    module.load_state_dict(synth.fill_state_dict(module.state_dict()))
    return module.cuda()


@pytest.fixture(scope="module")
def full_unet(gpu_device):
    from ldm_image_generator_amd.unet import UNet
    return formula(UNet())


def test_sampling_at_bench_shape_equals_its_rows(full_unet, gpu_device):
    from ldm_image_generator_amd.ddpm import DDPM
    from ldm_image_generator_amd.vae import Decoder
    dec = formula(Decoder())
    net = full_unet.eval()
    d = DDPM(model=net)
    x_t = torch.randn(256, 8, 32, 32, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        whole = dec(d.sample((256, 8, 32, 32), seed=100, num_steps=50, x_init=x_t, progress=False))
        assert torch.isfinite(whole).all()
        worst = 0.0
        for lo, hi in ((0, 2), (100, 103), (254, 256)):
            part = dec(d.sample((hi - lo, 8, 32, 32), seed=100, num_steps=50, x_init=x_t[lo:hi], progress=False))
            err = rel_l2(part.cpu(), whole[lo:hi].cpu())
            worst = max(worst, err)
            assert err < 5e-6, (lo, hi, err)
    print("bench-shape sampling: worst row-slice rel-L2 %.2e" % worst)
    del whole
    torch.cuda.empty_cache()


@pytest.mark.parametrize("prec,tol_loss,tol_grad", [("f32", 1e-5, 2e-4), ("bf16", 1e-5, 2e-4)])
def test_training_step_at_bench_shape_equals_mean_of_halves(gpu_device, prec, tol_loss, tol_grad):
    from ldm_image_generator_amd import train
    from ldm_image_generator_amd.train import L1LossFunction
    from ldm_image_generator_amd.unet import UNet
    net = formula(UNet()).train()
    train.set_precision(net, prec)
    gen = torch.Generator().manual_seed(31)
    x = torch.randn(128, 8, 64, 64, generator=gen).cuda()
    e = torch.randn(128, 8, 64, 64, generator=gen).cuda()
    t = torch.randint(1, 1000, (128,), generator=gen).cuda()

    def run(sl):
        for p in net.parameters():
            p.grad = None
        random.seed(55)
        loss = L1LossFunction.apply(net(x=x[sl], time=t[sl], condition=None), e[sl])
        loss.backward()
        out = float(loss), {k: (None if p.grad is None else p.grad.clone()) for k, p in net.named_parameters()}
        for p in net.parameters():
            p.grad = None
        return out

    l_all, g_all = run(slice(0, 128))
    l_a, g_a = run(slice(0, 64))
    l_b, g_b = run(slice(64, 128))
    train.set_precision(net, "f32")
    assert abs(l_all - 0.5 * (l_a + l_b)) < tol_loss * abs(l_all)
    worst, worst_k, used = 0.0, None, 0
    for k, ga in g_all.items():
        if ga is None:
            assert g_a[k] is None and g_b[k] is None, k
            continue
        used += 1
        assert torch.isfinite(ga).all(), k
        err = rel_l2(0.5 * (g_a[k] + g_b[k]), ga)
        if err > worst:
            worst, worst_k = err, k
    print("bench-shape training step (%s): %d used parameters, loss %.6f, worst half-batch rel-L2 %.2e (%s)" % (prec, used, l_all, worst, worst_k))
    assert used > 500
    assert worst < tol_grad, (worst_k, worst)
    del net, g_all, g_a, g_b
    torch.cuda.empty_cache()
