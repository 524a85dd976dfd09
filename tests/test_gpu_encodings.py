"""Encodings.proj1 in separable form (training step, one timestep per sample): kernels against torch on the same numbers, and
the whole Encodings forward / backward against autograd through the oracle's literal cat -> proj1 -> relu -> proj2."""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.mark.parametrize("B,HW,N,bf16", [(3, 20, 128, False), (4, 64, 256, True), (128, 16, 512, False), (2, 1024, 128, True), (5, 33, 64, False)])
def test_film_hidden_forward_and_backward_kernels(gpu_device, B, HW, N, bf16):
    from ldm_image_generator_amd import ops
    g = torch.Generator().manual_seed(B * HW + N)
    p_rows, t_rows = torch.randn(HW, N, generator=g), torch.randn(B, N, generator=g)
    ref = torch.relu(p_rows[None] + t_rows[:, None]).reshape(B * HW, N)
    hid = torch.empty(B * HW, N, device=gpu_device, dtype=BF if bf16 else torch.float32)
    ops.film_hidden(p_rows.cuda(), t_rows.cuda(), hid, B, HW, N)
    assert torch.equal(hid.cpu(), ref.to(BF) if bf16 else ref)
    dh = torch.randn(B * HW, N, generator=g)
    if bf16:
        dh = dh.to(BF)
    dp, dt = ops.film_hidden_bwd(dh.cuda(), hid, B, HW, N)
    dhm = (dh.double() * (hid.cpu().double() > 0)).reshape(B, HW, N)
    assert rel_l2(dp.cpu(), dhm.sum(0)) < 1e-6
    assert rel_l2(dt.cpu(), dhm.sum(1)) < 1e-6


@pytest.mark.parametrize("bf16", [False, True])
def test_encodings_separable_forward_backward_vs_autograd(gpu_device, bf16):
    """film and the gradients of proj1 / proj2 for per-sample timesteps vs torch autograd of the literal formulation
    (unet.py:18-21: cat[pe, te] -> proj1 -> relu -> proj2) on the oracle's sin / cos codes."""
    from ldm_image_generator_amd import synth, train
    from ldm_image_generator_amd.unet import Encodings, TimeContext
    from oracle import ldm_oracle as O
    C, H, W, B = 64, 8, 4, 4                      # M = 128 rows
    enc = Encodings(C)
    enc.load_state_dict(synth.fill_state_dict(enc.state_dict()))
    enc = enc.cuda()
    t = torch.tensor([7, 999, 431, 7])
    ctx = TimeContext(t.cuda(), B, gpu_device, dedupe=False)
    lc = train.LevelCodes(ctx, C, H, W)
    hid, film = train.encodings_forward(enc, lc, bf16)
    g = torch.Generator().manual_seed(1)
    dfilm = torch.randn(B * H * W, 2 * C, generator=g)
    grads = train.Grads()
    train.encodings_backward(enc, lc, hid, dfilm.cuda().to(BF) if bf16 else dfilm.cuda(), grads)
    # literal reference on the CPU
    sd = {k: v.detach().cpu().clone().requires_grad_() for k, v in enc.state_dict().items()}
    pe = O.positional_table(C, H, W).expand(B, C, H, W)
    te = O.time_table(C, t).reshape(B, C, 1, 1).expand(B, C, H, W)
    codes = torch.cat([pe, te], dim=1)                                                              # [B, 2C, H, W]
    rows = codes.permute(0, 2, 3, 1).reshape(-1, 2 * C)
    w1, w2 = sd["proj1.weight"].reshape(4 * C, 2 * C), sd["proj2.weight"].reshape(2 * C, 4 * C)
    ref = torch.relu(rows @ w1.t() + sd["proj1.bias"]) @ w2.t() + sd["proj2.bias"]
    ref.backward(dfilm)
    tol_f, tol_g = (2e-2, 3e-2) if bf16 else (1e-5, 1e-4)
    assert rel_l2(film.cpu(), ref.detach()) < tol_f
    for name, p in enc.named_parameters():
        assert rel_l2(grads.g[p].cpu().reshape(-1), sd[name].grad.reshape(-1)) < tol_g, name
