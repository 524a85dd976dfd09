"""Grouped 3x3 conv with bf16 operands (ldm_gconv3x3_bf16): exact-input test (bf16 values, fp64 evaluation of the same
numbers -> only fp32 accumulation error), forward with bias + residual and the data gradient through the flipped filter."""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.mark.parametrize("B,H,W,C", [(2, 8, 8, 64), (1, 5, 7, 32), (3, 16, 16, 128), (2, 4, 4, 1024), (1, 64, 64, 128), (2, 32, 32, 256), (1, 4, 8, 64)])
def test_gconv3x3_bf16_forward_and_data_gradient(gpu_device, B, H, W, C):
    from ldm_image_generator_amd import ops
    g = torch.Generator().manual_seed(B * H + C)
    x = torch.randn(B, C, H, W, generator=g).to(BF)
    wt = (torch.randn(C, 32, 3, 3, generator=g) / 17.0).to(BF)
    bias = torch.randn(C, generator=g)
    res = torch.randn(B, C, H, W, generator=g)
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), bias.double(), padding=1, groups=C // 32) + res.double()
    rows = x.permute(0, 2, 3, 1).reshape(-1, C).contiguous().cuda()
    res_rows = res.permute(0, 2, 3, 1).reshape(-1, C).contiguous().cuda()
    packed = wt.permute(0, 2, 3, 1).reshape(C, 288).contiguous().cuda()            # [co][tap][ci], like SwinBlock._conv_weight
    out = torch.full((B * H * W, C), float("nan"), device=gpu_device)
    ops.gconv3x3_bf16(rows, packed, bias.cuda(), res_rows, out, B, H, W, C)
    got = out.cpu().reshape(B, H, W, C).permute(0, 3, 1, 2)
    assert rel_l2(got, ref) < 1e-5
    # the LDS-tiled kernel (taken above where the shape allows) and the direct kernel run the same MFMA order: bit-identical
    old = ops.gconv3x3_bf16_tiled(0)
    out_direct = torch.full((B * H * W, C), float("nan"), device=gpu_device)
    ops.gconv3x3_bf16(rows, packed, bias.cuda(), res_rows, out_direct, B, H, W, C)
    ops.gconv3x3_bf16_tiled(old)
    assert torch.equal(out, out_direct)
    # data gradient: conv of dy with the spatially flipped, in/out-swapped filter, accumulated in place
    dy = torch.randn(B, C, H, W, generator=g).to(BF)
    xin = x.double().requires_grad_()
    torch.nn.functional.conv2d(xin, wt.double(), None, padding=1, groups=C // 32).backward(dy.double())
    gq = C // 32
    wrot = wt.reshape(gq, 32, 32, 3, 3).flip(3, 4).permute(0, 2, 3, 4, 1).reshape(C, 288).contiguous().cuda()
    acc0 = torch.randn(B * H * W, C, generator=g)
    acc = acc0.cuda().clone()
    ops.gconv3x3_bf16(dy.permute(0, 2, 3, 1).reshape(-1, C).contiguous().cuda(), wrot, None, acc, acc, B, H, W, C)
    want = xin.grad.permute(0, 2, 3, 1).reshape(-1, C) + acc0.double()
    assert rel_l2(acc.cpu(), want) < 1e-5
    # weight gradient (zero-padded index space, transposing LDS reads): dW[co][tap][ci] vs autograd on the same bf16 numbers
    wz = torch.zeros(C, 32, 3, 3, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x.double(), wz, None, padding=1, groups=C // 32).backward(dy.double())
    dw = ops.gconv3x3_wgrad_bf16(rows, dy.permute(0, 2, 3, 1).reshape(-1, C).contiguous().cuda(), B, H, W, C)
    got_w = dw.cpu().reshape(C, 3, 3, 32).permute(0, 3, 1, 2)
    assert rel_l2(got_w, wz.grad) < 1e-5
