"""The opt-in bf16 ("autocast") sampling / decode mode (ddpm.py:52,75: on a GPU the reference samples under 16-bit autocast).

Kernel level: exact-input tests -- operands are bf16 values, the reference is evaluated in fp64 and rounded once to bf16, so the
only admissible difference is the final rounding position (<= 1 bf16 ulp on a few elements: rel-L2 <= 4e-3, the bf16 rounding
noise level 2^-9 / sqrt(3)).  Bit-identity where two kernels run the same MFMA order (ring vs stream gate).

Model level: the bf16 mode against the REFERENCE's fp32 goldens (tests/golden/*.npz).  Stated tolerance = about twice what was
measured on MI355X (printed by the tests; measured: UNet forward 1.3e-3, 3-step latents 1.1e-3, 50-step latents 4.0e-4, decoded
image 2.5e-3 ... 2.8e-3): UNet forward rel-L2 <= 3e-3; 3-step latents <= 2.5e-3, 50-step latents <= 1.5e-3; decoded image <= 6e-3.  The default path stays exact fp32: `use_autocast` alone changes nothing.
"""
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import T, load_golden, rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
ULP_TOL = 4e-3


def bf(x):
    return x.to(BF)


def formula(module, gain=1.0):
    from ldm_image_generator_amd import synth
    module.load_state_dict(synth.fill_state_dict(module.state_dict(), gain=gain))
    return module.cuda()


@pytest.mark.parametrize("B,H,W,Cin,Cout,add", [(2, 16, 16, 64, 64, False), (1, 12, 20, 128, 128, True), (3, 9, 7, 64, 192, True),
                                                (4, 64, 64, 128, 512, True), (1, 32, 32, 256, 64, False)])
def test_conv3x3_bf16_implicit_gemm(gpu_device, B, H, W, Cin, Cout, add):
    from ldm_image_generator_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout)
    x = bf(torch.randn(B, Cin, H, W, generator=g))
    w = bf(torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5))
    bias = torch.randn(Cout, generator=g)
    skip = bf(torch.randn(B, Cout, H, W, generator=g)) if add else None
    y = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    y = F.leaky_relu(y, 0.01)
    if add:
        y = y + skip.double()
    rows = x.permute(0, 2, 3, 1).reshape(B * H * W, Cin).contiguous().cuda()
    wp = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().cuda()
    addr = skip.permute(0, 2, 3, 1).reshape(B * H * W, Cout).contiguous().cuda() if add else None
    out = torch.empty(B * H * W, Cout, device=gpu_device, dtype=BF)
    ops.gemm_bf16(rows, B * H * W, Cout, 9 * Cin, [wp], out, ldw=9 * Cin, biases=[bias.cuda()], act=ops.ACT_LRELU, slope=0.01, addend=addr,
                  a_mode=ops.A_CONV3X3, conv_hw=(H, W), cin=Cin)
    got = out.float().cpu().reshape(B, H, W, Cout).permute(0, 3, 1, 2)
    assert rel_l2(got, y) < ULP_TOL
    assert rel_l2(got, bf(y.float()).float()) < ULP_TOL


@pytest.mark.parametrize("M,C,nseg", [(512, 128, 3), (1024, 256, 3), (768, 128, 1)])
def test_gate_forward_hidden_only_ring_equals_stream(gpu_device, M, C, nseg):
    """ReGLU forward without saved pre-activations (sampling): the ring instance GF = 1 against the stream kernel, bit for bit."""
    from ldm_image_generator_amd import ops
    g = torch.Generator().manual_seed(M + C)
    x = bf(torch.randn(M, C, generator=g)).cuda()
    wa = [bf(torch.randn(C, C, generator=g) / C ** 0.5).cuda() for _ in range(nseg)]
    wb = [bf(torch.randn(C, C, generator=g) / C ** 0.5).cuda() for _ in range(nseg)]
    ba = [torch.randn(C, generator=g).cuda() for _ in range(nseg)]
    bb = [torch.randn(C, generator=g).cuda() for _ in range(nseg)]
    outs = []
    for ring in (0, 2, 3):
        old = ops.gemm_ring(ring)
        h = torch.empty(M, nseg * C, device=gpu_device, dtype=BF)
        ops.gemm_bf16_gate_fwd(x, M, nseg * C, C, wa, wb, h, biases_a=ba, biases_b=bb)
        ops.gemm_ring(old)
        outs.append(h)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    a = torch.cat([x.double().cpu() @ w.double().cpu().t() + b.double().cpu() for w, b in zip(wa, ba)], dim=1)
    b_ = torch.cat([x.double().cpu() @ w.double().cpu().t() + b.double().cpu() for w, b in zip(wb, bb)], dim=1)
    assert rel_l2(outs[0].float().cpu(), a * b_.clamp_min(0)) < ULP_TOL


@pytest.mark.parametrize("q16", [False, True])
@pytest.mark.parametrize("shift", [0, 3])
@pytest.mark.parametrize("hw", [(8, 8), (16, 16), (4, 4), (7, 9)])
def test_window_attention_bf16io(gpu_device, shift, hw, q16):
    """bf16 I/O around the fp32 attention core: with the same (bf16-representable) QKV, bias and mask values the context equals the
    fp32 kernel's, rounded once."""
    from ldm_image_generator_amd import ops
    B, C = 3, 64
    H, W = hw
    g = torch.Generator().manual_seed(H * 10 + W + shift)
    qkv = bf(torch.randn(B * H * W, 3 * C, generator=g)).cuda()
    bias = bf(torch.randn(3 * C, generator=g)).float().cuda()                     # padded tokens take bf16(bias) in the bf16-QKV form
    xf = bf(torch.randn(B * H * W, C, generator=g)).cuda()
    ref = torch.empty(B * H * W, C, device=gpu_device)
    ops.window_attention(qkv.float(), bias, xf.float(), ref, B, H, W, C, 6, shift)
    out = torch.empty(B * H * W, C, device=gpu_device, dtype=BF)
    ops.window_attention_bf16io(qkv if q16 else qkv.float(), bias, xf, out, B, H, W, C, 6, shift)
    if not q16:
        assert torch.equal(out, bf(ref))
        return
    # bf16 QKV: both products on the bf16 matrix cores, P rounded once to bf16 -> within bf16 rounding of the fp32 core's context ...
    assert rel_l2(out.float().cpu(), ref.cpu()) < ULP_TOL
    # ... and the fp32 core behind the same entry point (A/B switch) is exact
    old = ops.window_attention_bf16_core(0)
    out0 = torch.empty_like(out)
    ops.window_attention_bf16io(qkv, bias, xf, out0, B, H, W, C, 6, shift)
    ops.window_attention_bf16_core(old)
    assert torch.equal(out0, bf(ref))


def test_bf16_helpers(gpu_device):
    from ldm_image_generator_amd import ops
    g = torch.Generator().manual_seed(3)
    B, H, W, C = 2, 6, 10, 64
    # depth to space
    quad = bf(torch.randn(B * H * W, 4 * C, generator=g))
    fine = ops.depth_to_space2_bf16(quad.cuda(), B, H, W, C).cpu()
    ref = quad.reshape(B, H, W, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B * 2 * H * 2 * W, C)
    assert torch.equal(fine, ref)
    # stem with bf16 rows out
    x = torch.randn(B, 8, H, W, generator=g)
    w = torch.randn(C, 8, generator=g) / 8 ** 0.5
    b = torch.randn(C, generator=g)
    rows32 = torch.empty(B * H * W, C, device=gpu_device)
    ops.stem_nchw(x.cuda(), w.cuda(), b.cuda(), rows32, B, 8, H * W, C)
    rows16 = torch.empty(B * H * W, C, device=gpu_device, dtype=BF)
    ops.stem_nchw_bf16(x.cuda(), w.cuda(), b.cuda(), rows16, B, 8, H * W, C)
    assert torch.equal(rows16, bf(rows32))
    # rgb head on bf16 rows == the fp32 kernel on the widened rows
    r16 = bf(torch.randn(B * H * W, C, generator=g)).cuda()
    wr = (torch.randn(3, C, generator=g) / C ** 0.5).cuda()
    br = torch.randn(3, generator=g).cuda()
    prev = torch.randn(B, 3, H // 2, W // 2, generator=g).cuda()
    o32 = torch.empty(B, 3, H, W, device=gpu_device)
    o16 = torch.empty(B, 3, H, W, device=gpu_device)
    ops.rgb_head(r16.float(), wr, br, prev, o32, B, H, W, C)
    ops.rgb_head_bf16(r16, wr, br, prev, o16, B, H, W, C)
    assert rel_l2(o16.cpu(), o32.cpu()) < 1e-6
    # nearest x2 + skip
    coarse = torch.randn(B * H * W, C, generator=g)
    skip = torch.randn(B * 4 * H * W, C, generator=g)
    out = torch.empty(B * 4 * H * W, C, device=gpu_device)
    ops.up2_add(coarse.cuda(), skip.cuda(), out, B, H, W, C)
    ref = coarse.reshape(B, H, 1, W, 1, C).expand(B, H, 2, W, 2, C).reshape(B * 4 * H * W, C) + skip
    assert torch.equal(out.cpu(), ref)
    # average pool with one rounding
    xin = torch.randn(B * H * W, C, generator=g).cuda()
    p32 = torch.empty(B * (H // 2) * (W // 2), C, device=gpu_device)
    ops.avgpool2(xin, p32, B, H, W, C)
    p16 = torch.empty(B * (H // 2) * (W // 2), C, device=gpu_device, dtype=BF)
    ops.avgpool2_bf16(xin, p16, B, H, W, C)
    assert torch.equal(p16, bf(p32))


def test_use_autocast_alone_changes_nothing(gpu_device):
    """Without the opt-in the flag is inert: the default path is exact fp32 (bit-identical with and without use_autocast)."""
    from ldm_image_generator_amd.ddpm import DDPM
    from ldm_image_generator_amd.unet import UNet
    net = formula(UNet(input_channels=8, stages=[1, 2], channels=[64, 128])).eval()
    d = DDPM(model=net)
    xT = torch.randn(2, 8, 16, 16, generator=torch.Generator().manual_seed(0))
    a = d.sample((2, 8, 16, 16), seed=1, num_steps=3, x_init=xT, progress=False, use_autocast=True)
    b = d.sample((2, 8, 16, 16), seed=1, num_steps=3, x_init=xT, progress=False, use_autocast=False)
    assert torch.equal(a, b)


def test_bf16_sampling_small_net_vs_fp32_path(gpu_device):
    from ldm_image_generator_amd import autocast
    from ldm_image_generator_amd.ddpm import DDPM
    from ldm_image_generator_amd.unet import UNet
    net = formula(UNet(input_channels=8, stages=[1, 2], channels=[64, 128]))
    d = DDPM(model=net)
    xT = torch.randn(4, 8, 16, 16, generator=torch.Generator().manual_seed(0))
    for mode in ("eval", "train"):
        net.train(mode == "train")
        autocast.set_autocast_dtype(net, None)
        ref = d.sample((4, 8, 16, 16), seed=3, num_steps=5, x_init=xT, progress=False)
        autocast.set_autocast_dtype(net, torch.bfloat16)
        got = d.sample((4, 8, 16, 16), seed=3, num_steps=5, x_init=xT, progress=False)
        off = d.sample((4, 8, 16, 16), seed=3, num_steps=5, x_init=xT, progress=False, use_autocast=False)
        autocast.set_autocast_dtype(net, None)
        assert torch.equal(off, ref)                              # opted in, but use_autocast=False: exact fp32
        err = rel_l2(got.cpu(), ref.cpu())
        print("bf16 sampling, small net, %s mode, 5 steps: rel-L2 %.3e vs the fp32 path" % (mode, err))
        assert err < 1.5e-3


@pytest.fixture(scope="module")
def full_unet(gpu_device):
    from ldm_image_generator_amd.unet import UNet
    return formula(UNet())


def test_bf16_unet_forward_full_size_vs_reference(full_unet):
    from ldm_image_generator_amd import autocast
    g = load_golden("unet_full")
    x, t = T(g["x"]).cuda(), T(g["t"]).cuda()
    net = full_unet
    autocast.set_autocast_dtype(net, torch.bfloat16)
    try:
        with torch.no_grad():
            for mode, key in (("eval", "y_eval"), ("train", "y_train_0")):
                net.train(mode == "train")
                net._autocast_now = True
                random.seed(0)
                y = net(x, t)
                net._autocast_now = False
                err = rel_l2(y.cpu(), T(g[key]))
                print("bf16 UNet forward, full size, %s: rel-L2 %.3e vs the reference's fp32 output" % (mode, err))
                assert err < 3e-3
    finally:
        net._autocast_now = False
        autocast.set_autocast_dtype(net, None)


def test_bf16_ddim_sample_full_size_vs_reference(full_unet):
    from ldm_image_generator_amd import autocast
    from ldm_image_generator_amd.ddpm import DDPM
    g = load_golden("sample_full")
    net = full_unet
    d = DDPM(model=net)
    autocast.set_autocast_dtype(net, torch.bfloat16)
    try:
        net.train()
        x0 = d.sample((1, 8, 32, 32), seed=0, num_steps=50, x_init=T(g["xT"]), progress=False).cpu()
        e50 = rel_l2(x0, T(g["x0_train_50"]))
        net.eval()
        x0 = d.sample((1, 8, 32, 32), seed=0, num_steps=3, x_init=T(g["xT"]), progress=False).cpu()
        e3 = rel_l2(x0, T(g["x0_eval_3"]))
    finally:
        autocast.set_autocast_dtype(net, None)
    print("bf16 DDIM sampling, full size: 50 steps (train mode) rel-L2 %.3e, 3 steps (eval) %.3e vs the reference's fp32 latents" % (e50, e3))
    assert e3 < 2.5e-3 and e50 < 1.5e-3


def test_bf16_decoder_full_size_vs_reference(gpu_device):
    from ldm_image_generator_amd import autocast
    from ldm_image_generator_amd.vae import Decoder
    g = load_golden("decoder_full")
    dec = formula(Decoder())
    autocast.set_compute_dtype(dec, torch.bfloat16)
    with torch.no_grad():
        y = dec(T(g["z"]).cuda()).cpu()
    e_sub = rel_l2(y[:, :, ::4, ::4], T(g["y_sub"]))
    e_rows = rel_l2(y[:, :, 100:104, :], T(g["y_rows"]))
    print("bf16 decode, full size: rel-L2 %.3e (subsampled image), %.3e (rows 100-103) vs the reference's fp32 image" % (e_sub, e_rows))
    assert e_sub < 6e-3 and e_rows < 6e-3
    autocast.set_compute_dtype(dec, None)
    with torch.no_grad():
        y32 = dec(T(g["z"]).cuda()).cpu()
    assert rel_l2(y32[:, :, ::4, ::4], T(g["y_sub"])) < 1e-5                 # switching back restores the exact path


def test_bf16_decoder_tiny_vs_fp32_path(gpu_device):
    from ldm_image_generator_amd import autocast
    from ldm_image_generator_amd.vae import Decoder
    dec = formula(Decoder(channels=[128, 64, 64], stages=[1, 2, 1]))
    z = torch.randn(3, 8, 6, 10, generator=torch.Generator().manual_seed(2)).cuda()
    with torch.no_grad():
        ref = dec(z)
        autocast.set_compute_dtype(dec, torch.bfloat16)
        got = dec(z)
    err = rel_l2(got.cpu(), ref.cpu())
    print("bf16 decode, tiny net: rel-L2 %.3e vs the fp32 path" % err)
    assert err < 6e-3
