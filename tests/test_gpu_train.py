"""Training step on the HIP path (forward with saved activations + hand-written backward) against
(a) the gradient norms the REFERENCE produced through torch autograd (tests/golden/loss_tiny.npz) and
(b) autograd through the CPU oracle on fresh inputs.  fp32 tolerances: loss rel 1e-5, gradients rel-L2 1e-4."""
import random

import numpy as np
import pytest
import torch

from conftest import T, load_golden, rel_l2
from oracle import ldm_oracle as O

pytestmark = pytest.mark.gpu

TINY = dict(input_channels=8, stages=[1, 2, 3, 2], channels=[32, 64, 96, 128])


def formula(module, gain=1.0):
    from ldm_image_generator_amd import synth
    module.load_state_dict(synth.fill_state_dict(module.state_dict(), gain=gain))
    return module.cuda()


@pytest.mark.parametrize("M,N,K,S", [(64, 128, 128, 1), (4096, 256, 128, 4), (2048, 128, 384, 2), (96, 384, 256, 3), (4096, 128, 128, 7), (6272, 128, 256, 22)])
def test_gemm_tn_weight_gradient_kernel(gpu_device, M, N, K, S):
    """dW = dY^T X straight from the row-major operands (no transposed copies): every split plane and their sum.  Splits that do not
    divide M: rows per split = M / S rounded up to 32, the last split shorter."""
    from ldm_image_generator_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    dy = torch.randn(M, N, generator=g).cuda()
    x = torch.randn(M, K, generator=g).cuda()
    parts = torch.empty(S, N, K, device=gpu_device)
    cs = torch.empty(S, N, device=gpu_device)
    ops.gemm_tn(dy, x, parts, M, N, K, S, colsum=cs)
    assert rel_l2(cs.sum(0).double().cpu(), dy.double().sum(0).cpu()) < 1e-5
    ms = ((M + S - 1) // S + 31) // 32 * 32
    for s_ in range(S):
        ref = dy[s_ * ms:(s_ + 1) * ms].double().t() @ x[s_ * ms:(s_ + 1) * ms].double()
        assert rel_l2(parts[s_].double().cpu(), ref.cpu()) < 1e-5, s_
    out = torch.empty(N, K, device=gpu_device)
    ops.reduce_partials(parts, S, N * K, out)
    assert rel_l2(out.double().cpu(), (dy.double().t() @ x.double()).cpu()) < 1e-5
    # leading dimension wider than N (the first N columns of a wider matrix)
    wide = torch.randn(M, N + 128, generator=g).cuda()
    ops.gemm_tn(wide, x, parts, M, N, K, S, lda=N + 128)
    ops.reduce_partials(parts, S, N * K, out)
    assert rel_l2(out.double().cpu(), (wide[:, :N].double().t() @ x.double()).cpu()) < 1e-5


@pytest.mark.parametrize("S,n", [(1, 4096), (3, 1000), (4, 256), (5, 49152), (7, 260), (128, 49152), (13, 3 * 1024 * 1024), (64, 37)])
def test_reduce_partials_fixed_order_sum(gpu_device, S, n):
    """Sum of S partial planes: vectorised kernel (n % 4 == 0, S >= 4: four quarter sums in plane order) and the scalar one;
    deterministic (two runs bit-equal), fp32-accurate against the float64 sum."""
    from ldm_image_generator_amd import ops
    parts = torch.randn(S, n, generator=torch.Generator().manual_seed(S * 7 + n)).cuda()
    out = torch.full((n,), float("nan"), device=gpu_device)
    again = torch.full((n,), float("nan"), device=gpu_device)
    ops.reduce_partials(parts, S, n, out)
    ops.reduce_partials(parts, S, n, again)
    assert torch.equal(out, again)
    ref = parts.double().sum(0)
    assert float((out.double() - ref).abs().max()) <= 4e-7 * float(parts.abs().double().sum(0).max())


@pytest.mark.parametrize("S,na,nb", [(1, 64, 8), (2, 4096, 128), (16, 128 * 384, 384), (128, 1000 * 4, 12), (7, 256, 260)])
def test_reduce_partials_pair_equals_two_single_launches(gpu_device, S, na, nb):
    """ldm_reduce_partials_pair_f32 (a weight gradient's planes and its bias gradient's in one launch) == two single launches, bit for bit."""
    from ldm_image_generator_amd import ops
    g = torch.Generator().manual_seed(S + na)
    pa, pb = torch.randn(S, na, generator=g).cuda(), torch.randn(S, nb, generator=g).cuda()
    oa, ob = torch.full((na,), float("nan"), device=gpu_device), torch.full((nb,), float("nan"), device=gpu_device)
    ops.reduce_partials_pair(pa, na, oa, pb, nb, ob, S)
    ra, rb = torch.empty(na, device=gpu_device), torch.empty(nb, device=gpu_device)
    ops.reduce_partials(pa, S, na, ra)
    ops.reduce_partials(pb, S, nb, rb)
    assert torch.equal(oa, ra) and torch.equal(ob, rb)


@pytest.mark.parametrize("S,rows,row_len,seg", [(2, 128, 384, 128), (5, 96, 48, 16), (16, 1024, 3072, 1024), (3, 8, 24, 4)])
def test_reduce_partials_pair_column_blocks(gpu_device, S, rows, row_len, seg):
    """seg_len_a: the summed [rows, row_len] matrix comes back as contiguous [row_len / seg][rows][seg] column blocks (the c-weight
    gradients of a block's three ReGLUs out of one weight-gradient GEMM), the same bits as slicing the plain sum."""
    from ldm_image_generator_amd import ops
    g = torch.Generator().manual_seed(S + rows)
    na, nb = rows * row_len, rows
    pa, pb = torch.randn(S, na, generator=g).cuda(), torch.randn(S, nb, generator=g).cuda()
    plain, ob = torch.empty(rows, row_len, device=gpu_device), torch.empty(nb, device=gpu_device)
    ops.reduce_partials_pair(pa, na, plain, pb, nb, ob, S)
    blocks, ob2 = torch.full((row_len // seg, rows, seg), float("nan"), device=gpu_device), torch.empty(nb, device=gpu_device)
    ops.reduce_partials_pair(pa, na, blocks, pb, nb, ob2, S, row_len_a=row_len, seg_len_a=seg)
    assert torch.equal(ob, ob2)
    for e in range(row_len // seg):
        assert torch.equal(blocks[e], plain[:, e * seg:(e + 1) * seg])
    with pytest.raises(RuntimeError):
        ops.reduce_partials_pair(pa, na, blocks, pb, nb, ob2, S, row_len_a=row_len, seg_len_a=seg + 2)


@pytest.mark.parametrize("B,H,W,C,S", [(2, 8, 8, 64, 1), (4, 16, 16, 128, 2), (32, 4, 4, 256, 4), (1, 6, 64, 32, 1), (3, 32, 32, 64, 8)])
def test_grouped_conv_weight_gradient_kernel(gpu_device, B, H, W, C, S):
    """dW of the 32-per-group 3x3 conv from the row-major activations (no transposed im2col) vs autograd."""
    from ldm_image_generator_amd import ops
    g = torch.Generator().manual_seed(B * H + C)
    x = torch.randn(B, C, H, W, generator=g)
    dy = torch.randn(B, C, H, W, generator=g)
    wt = torch.zeros(C, 32, 3, 3, requires_grad=True)
    torch.nn.functional.conv2d(x, wt, padding=1, groups=C // 32).backward(dy)
    rows = x.permute(0, 2, 3, 1).reshape(-1, C).contiguous().cuda()
    drows = dy.permute(0, 2, 3, 1).reshape(-1, C).contiguous().cuda()
    planes = torch.full((4 * S, C, 288), float("nan"), device=gpu_device)
    ops.gconv3x3_wgrad(rows, drows, planes, B, H, W, C, S)
    dw = torch.empty(C, 288, device=gpu_device)
    ops.reduce_partials(planes, 4 * S, C * 288, dw)
    got = dw.cpu().reshape(C, 3, 3, 32).permute(0, 3, 1, 2)
    assert rel_l2(got, wt.grad) < 1e-5


def test_backward_kernels_against_autograd(gpu_device):
    from ldm_image_generator_amd import ops
    g = torch.Generator().manual_seed(0)
    # ChannelNorm + FiLM backward
    B, HW, C = 3, 16, 96
    x = (torch.randn(B * HW, C, generator=g) * 2 + 0.3).requires_grad_()
    film = torch.randn(2 * HW, 2 * C, generator=g).requires_grad_()
    slot = torch.tensor([1, 0, 1])
    dxf, dres = torch.randn(B * HW, C, generator=g), torch.randn(B * HW, C, generator=g)
    xn = O.channel_norm(x.reshape(B, HW, C).permute(0, 2, 1).reshape(B, C, HW, 1)).reshape(B, C, HW).permute(0, 2, 1)
    fs = film.reshape(2, HW, 2 * C)[slot]
    (xn * fs[:, :, :C] + fs[:, :, C:]).backward(dxf.reshape(B, HW, C))
    dx = torch.empty(B * HW, C, device=gpu_device)
    dfilm = torch.zeros(2 * HW, 2 * C, device=gpu_device)
    ops.channelnorm_film_bwd(x.detach().cuda(), film.detach().cuda(), slot.int().cuda(), dxf.cuda(), dres.cuda(), dx, dfilm, B, HW, C)
    assert rel_l2(dx.cpu(), x.grad + dres) < 1e-5
    assert rel_l2(dfilm.cpu(), film.grad) < 1e-5
    # one slot per sample (the training path): plain stores into an UNINITIALISED buffer, no atomics
    film3 = torch.randn(B * HW, 2 * C, generator=g).requires_grad_()
    x.grad = None
    f3 = film3.reshape(B, HW, 2 * C)
    (xn.detach() * 0 + O.channel_norm(x.reshape(B, HW, C).permute(0, 2, 1).reshape(B, C, HW, 1)).reshape(B, C, HW).permute(0, 2, 1)
     * f3[:, :, :C] + f3[:, :, C:]).backward(dxf.reshape(B, HW, C))
    dx3 = torch.empty(B * HW, C, device=gpu_device)
    dfilm3 = torch.full((B * HW, 2 * C), float("nan"), device=gpu_device)
    ops.channelnorm_film_bwd(x.detach().cuda(), film3.detach().cuda(), torch.arange(B, dtype=torch.int32).cuda(), dxf.cuda(), dres.cuda(), dx3,
                             dfilm3, B, HW, C, unique_slots=True)
    assert rel_l2(dx3.cpu(), x.grad + dres) < 1e-5
    assert rel_l2(dfilm3.cpu(), film3.grad) < 1e-5
    # gate / relu / pooling / colsum / l1
    a, b, dh = (torch.randn(64, 96, generator=g) for _ in range(3))
    da, db = torch.empty(64, 96, device=gpu_device), torch.empty(64, 96, device=gpu_device)
    ops.gate_bwd(dh.cuda(), a.cuda(), b.cuda(), da, db)
    assert torch.allclose(da.cpu(), dh * torch.relu(b)) and torch.allclose(db.cpu(), dh * a * (b > 0))
    hid = torch.empty(64, 96, device=gpu_device)
    assert torch.allclose(ops.gate_fwd(a.cuda(), b.cuda(), hid).cpu(), a * torch.relu(b))
    xs = torch.randn(1500, 70, generator=g)
    assert rel_l2(ops.colsum(xs.cuda(), 1500, 70).cpu(), xs.double().sum(0)) < 1e-6
    hi = torch.randn(2, 6, 8, 32, generator=g)
    lo = torch.empty(2 * 3 * 4, 32, device=gpu_device)
    ops.sumpool2(hi.reshape(-1, 32).cuda(), lo, 2, 6, 8, 32)
    ref = torch.nn.functional.avg_pool2d(hi.permute(0, 3, 1, 2), 2) * 4
    assert rel_l2(lo.cpu().reshape(2, 3, 4, 32).permute(0, 3, 1, 2), ref) < 1e-6
    base = torch.randn(2 * 6 * 8, 32, generator=g)
    acc = base.cuda().clone()
    ops.avgpool2_bwd(lo, acc, 2, 6, 8, 32, True)
    up = torch.nn.functional.interpolate(lo.cpu().reshape(2, 3, 4, 32).permute(0, 3, 1, 2), scale_factor=2) * 0.25
    assert rel_l2(acc.cpu(), base + up.permute(0, 2, 3, 1).reshape(-1, 32)) < 1e-6
    p, q = torch.randn(4, 8, 16, 16, generator=g), torch.randn(4, 8, 16, 16, generator=g)
    loss = torch.empty(1, device=gpu_device)
    ops.l1_loss(p.cuda(), q.cuda(), loss)
    assert abs(float(loss) - float((p - q).abs().mean())) < 1e-6


@pytest.mark.parametrize("B,Cin,HW,C0", [(2, 8, 35, 128), (3, 3, 100, 32), (1, 8, 4096, 96), (2, 12, 300, 64), (64, 8, 4096, 128), (2, 8, 64, 320)])
def test_stem_weight_gradient_kernel(gpu_device, B, Cin, HW, C0):
    """dw[n, ci] = sum_m dy[m, n] x[b, ci, p] (the gradient of the 1x1 stem from NCHW, unet.py:77): the thread-per-output-channel
    kernel (C0 <= 256) and the generic one (C0 = 320) against fp64."""
    from ldm_image_generator_amd import ops
    g = torch.Generator().manual_seed(B * 7 + C0)
    x = torch.randn(B, Cin, HW, generator=g)
    dy = torch.randn(B * HW, C0, generator=g)
    dw = torch.full((C0, Cin), float("nan"), device=gpu_device)
    ops.stem_bwd(x.cuda(), dy.cuda(), dw, B, Cin, HW, C0)
    ref = torch.einsum("bpn,bcp->nc", dy.double().reshape(B, HW, C0), x.double())
    assert rel_l2(dw.cpu().double(), ref) < 1e-5


@pytest.mark.parametrize("B,C0,HW,Cin", [(2, 128, 1000, 8), (1, 32, 70, 3), (4, 64, 40000, 8), (2, 512, 64, 8), (1, 96, 33, 16)])
def test_head_backward_kernel(gpu_device, B, C0, HW, Cin):
    """out[b, co, p] = sum_c x[m, c] w[c, co] + bias (unet.py:78 as a rows -> NCHW kernel): dx, dw, db against fp64; 2 500 tiles on
    2 048 blocks (grid-stride accumulation) and a 4 096-weight head (the per-tile atomics tail) included."""
    from ldm_image_generator_amd import ops
    g = torch.Generator().manual_seed(B * 11 + C0)
    M = B * HW
    x = torch.randn(M, C0, generator=g)
    w = torch.randn(C0, Cin, generator=g) * C0 ** -0.5
    dout = torch.randn(B, Cin, HW, generator=g)
    dx = torch.full((M, C0), float("nan"), device=gpu_device)
    dw = torch.full((C0, Cin), float("nan"), device=gpu_device)
    db = torch.full((Cin,), float("nan"), device=gpu_device)
    ops.head_bwd(x.cuda(), w.cuda(), dout.cuda(), dx, dw, db, B, C0, HW, Cin)
    drows = dout.double().permute(0, 2, 1).reshape(M, Cin)
    assert rel_l2(dx.cpu().double(), drows @ w.double().t()) < 1e-5
    assert rel_l2(dw.cpu().double(), x.double().t() @ drows) < 1e-5
    assert rel_l2(db.cpu().double(), drows.sum(0)) < 1e-5


@pytest.mark.parametrize("B,H,W,C,shift", [(2, 8, 8, 64, 0), (2, 8, 8, 64, 3), (1, 20, 14, 128, 3), (3, 16, 16, 32, 0), (2, 4, 4, 64, 0),
                                            (2, 5, 5, 32, 0), (2, 3, 2, 32, 0), (1, 6, 6, 96, 0), (1, 12, 12, 64, 3)])
def test_window_attention_backward_mfma_kernel_matches_scalar_kernel(gpu_device, B, H, W, C, shift):
    """The MFMA backward (16x16x4 fp32 products, both orientations of the score matrix) against the scalar kernel on the same
    inputs: windows with zero-padded tokens, the float-mask quirk (shift 3), global attention at 16 / 25 / 6 / 36 tokens."""
    from ldm_image_generator_amd import ops
    g = torch.Generator().manual_seed(H * 31 + W + shift)
    m = B * H * W
    qkv = (torch.randn(m, 3 * C, generator=g) * 1.5).cuda()
    bias = torch.randn(3 * C, generator=g).cuda()
    xf = torch.randn(m, C, generator=g).cuda()
    dctx = torch.randn(m, C, generator=g).cuda()
    outs = []
    for mode in (0, 1):
        old = ops.window_attention_bwd_mfma(mode)
        try:
            dqkv = torch.full((m, 3 * C), float("nan"), device=gpu_device)
            dpad = torch.full((3 * C,), float("nan"), device=gpu_device)
            ops.window_attention_bwd(qkv, bias, xf, dctx, dqkv, dpad, B, H, W, C, 6, shift)
        finally:
            ops.window_attention_bwd_mfma(old)
        outs.append((dqkv, dpad))
    assert torch.isfinite(outs[1][0]).all() and torch.isfinite(outs[1][1]).all()
    for part in range(3):
        a, b_ = outs[0][0][:, part * C:(part + 1) * C], outs[1][0][:, part * C:(part + 1) * C]
        assert rel_l2(b_.cpu().double(), a.cpu().double()) < 1e-5, part
    assert float((outs[1][1] - outs[0][1]).abs().max()) <= 1e-5 * max(1.0, float(outs[0][1].abs().max()))


@pytest.mark.parametrize("shift,hw", [(0, 8), (3, 8), (3, 16), (0, 4), (0, 5), (3, 10)])
def test_window_attention_backward(gpu_device, shift, hw):
    """Gradients of the attention block (in-proj, windows incl. padded tokens and the float-mask quirk, out-proj)
    against autograd through the oracle.  The float "mask" is detached in the reference (attention.py:76-81 runs
    under no_grad), the oracle mirrors that."""
    from ldm_image_generator_amd import synth, train
    from ldm_image_generator_amd.attention import WindowAttention
    C = 64
    wa = formula(WindowAttention(C, n_heads=2, window_size=6, shift=shift), gain=2.0)
    sd = {k: v.clone().requires_grad_() for k, v in synth.fill_state_dict(wa.state_dict(), gain=2.0).items()}
    g = torch.Generator().manual_seed(hw + shift)
    x = torch.randn(2, C, hw, hw, generator=g).requires_grad_()
    dy = torch.randn(2, C, hw, hw, generator=g)
    y = O.window_attention(sd, "", x, 6, shift)
    y.backward(dy)
    # HIP path, driven like train.block_backward drives it
    from ldm_image_generator_amd import ops
    from ldm_image_generator_amd.modules import from_rows, to_rows
    rows, shape = to_rows(x.detach().cuda())
    drows, _ = to_rows(dy.cuda())
    b, h, w = shape
    m = rows.shape[0]
    att = wa.attention
    qkv = torch.empty(m, 3 * C, device=gpu_device)
    ops.gemm(rows, m, 3 * C, C, [att.in_proj_weight.detach()], qkv, biases=[att.in_proj_bias.detach()])
    dctx = torch.empty(m, C, device=gpu_device)
    ops.gemm(drows, m, C, C, [train._T(att.out_proj.weight.detach())], dctx)
    dqkv = torch.empty(m, 3 * C, device=gpu_device)
    dpad = torch.empty(3 * C, device=gpu_device)
    ops.window_attention_bwd(qkv, att.in_proj_bias.detach(), rows, dctx, dqkv, dpad, b, h, w, C, 6, shift)
    dx = torch.empty(m, C, device=gpu_device)
    ops.gemm(dqkv, m, C, 3 * C, [train._T(att.in_proj_weight.detach())], dx)
    assert rel_l2(from_rows(dx, shape).cpu(), x.grad) < 1e-4
    db_in = ops.add_(ops.colsum(dqkv, m, 3 * C), dpad)
    assert rel_l2(db_in.cpu(), sd["attention.in_proj_bias"].grad) < 1e-4
    if m % 32 == 0:
        dw_in = train.grad_weight(train._T(dqkv), train._T(rows), m)
        assert rel_l2(dw_in.cpu(), sd["attention.in_proj_weight"].grad) < 1e-4


def test_training_step_matches_reference_gradients(gpu_device):
    """loss.backward() through UNetFunction vs the per-parameter gradient norms of the reference's own autograd."""
    from ldm_image_generator_amd.ddpm import DDPM
    from ldm_image_generator_amd.train import L1LossFunction
    from ldm_image_generator_amd.unet import UNet
    from ldm_image_generator_amd import ops
    g = load_golden("loss_tiny")
    net = formula(UNet(**TINY)).train()
    d = DDPM(model=net)
    x, t, e = T(g["x"]).cuda(), T(g["t"]), T(g["e"]).cuda()
    ab = d.alpha_bar[t]
    xt = torch.empty_like(x)
    ops.qsample(x, e, torch.sqrt(ab).cuda(), torch.sqrt(1 - ab).cuda(), xt)
    random.seed(3)
    e_theta = net(x=xt, time=t.cuda(), condition=None)
    loss = L1LossFunction.apply(e_theta, e)
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    names = [str(n) for n in g["grad_names"]]
    norms = dict(zip(names, g["grad_norms"]))
    worst = 0.0
    for k, p in net.named_parameters():
        ref = norms[k]
        if ref < 0:
            assert p.grad is None, k                     # unused expert / skipped block / dead cross-attention
            continue
        assert p.grad is not None, k
        got = float(p.grad.double().norm())
        err = abs(got - ref) / max(ref, 1e-12)
        worst = max(worst, err)
        assert err < 2e-4, (k, got, ref)
    assert rel_l2(net.encoder_first.weight.grad.cpu(), T(g["grad_encoder_first_weight"])) < 1e-4
    assert rel_l2(net.decoder_last.weight.grad.cpu(), T(g["grad_decoder_last_weight"])) < 1e-4


@pytest.mark.parametrize("M,C,ring", [(65536, 256, 2), (4096, 128, 2), (1024, 64, 1), (768, 96, 1)])
def test_gemm_gate_fwd_equals_three_launches(gpu_device, M, C, ring):
    """ldm_gemm_f32_gate_fwd (hidden + both pre-activations of three ReGLUs in one launch of the ring kernel's gated instance) ==
    two plain GEMMs + gate_fwd, bit for bit; shapes the ring kernel does not take fall back to exactly those three launches."""
    from ldm_image_generator_amd import ops
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g).cuda()
    wa = [(torch.randn(C, C, generator=g) * C ** -0.5).cuda() for _ in range(3)]
    wb = [(torch.randn(C, C, generator=g) * C ** -0.5).cuda() for _ in range(3)]
    ba = [torch.randn(C, generator=g).cuda() for _ in range(3)]
    bb = [torch.randn(C, generator=g).cuda() for _ in range(3)]
    ref_a, ref_b, ref_h = (torch.empty(M, 3 * C, device=gpu_device) for _ in range(3))
    old = ops.gemm_ring(ring)
    try:
        ops.gemm(x, M, 3 * C, C, wa, ref_a, biases=ba)
        ops.gemm(x, M, 3 * C, C, wb, ref_b, biases=bb)
        ops.gate_fwd(ref_a, ref_b, ref_h)
        a_pre, b_pre, hid = (torch.full((M, 3 * C), float("nan"), device=gpu_device) for _ in range(3))
        ops.gemm_gate_fwd(x, M, 3 * C, C, wa, wb, hid, a_pre, b_pre, biases_a=ba, biases_b=bb)
    finally:
        ops.gemm_ring(old)
    assert torch.equal(a_pre, ref_a) and torch.equal(b_pre, ref_b) and torch.equal(hid, ref_h)
    ref64 = (x.double() @ torch.cat(wa).double().t() + torch.cat(ba).double()) * torch.relu(x.double() @ torch.cat(wb).double().t() + torch.cat(bb).double())
    assert rel_l2(hid.double().cpu(), ref64.cpu()) < 1e-5


@pytest.mark.parametrize("M,C,ring", [(65536, 256, 2), (4096, 128, 2), (1024, 64, 1)])
def test_gemm_gate_bwd_equals_two_launches(gpu_device, M, C, ring):
    """ldm_gemm_f32_gate_bwd (da, db formed in the epilogue of dh = dy . Wc) == plain GEMM + gate_bwd, bit for bit."""
    from ldm_image_generator_amd import ops
    g = torch.Generator().manual_seed(M + 3 * C)
    dy = torch.randn(M, C, generator=g).cuda()
    wt = [(torch.randn(C, C, generator=g) * C ** -0.5).cuda() for _ in range(3)]          # transposed c-weights [F, C] per expert
    a_pre, b_pre = torch.randn(M, 3 * C, generator=g).cuda(), torch.randn(M, 3 * C, generator=g).cuda()
    old = ops.gemm_ring(ring)
    try:
        dh = torch.empty(M, 3 * C, device=gpu_device)
        ops.gemm(dy, M, 3 * C, C, wt, dh)
        ref_a, ref_b = torch.empty_like(dh), torch.empty_like(dh)
        ops.gate_bwd(dh, a_pre, b_pre, ref_a, ref_b)
        da, db = torch.full_like(dh, float("nan")), torch.full_like(dh, float("nan"))
        ops.gemm_gate_bwd(dy, M, 3 * C, C, wt, a_pre, b_pre, da, db)
    finally:
        ops.gemm_ring(old)
    assert torch.equal(da, ref_a) and torch.equal(db, ref_b)


def test_training_step_stem_size_2(gpu_device):
    """stem_size = 2: loss and every parameter gradient of one calculate_loss backward vs the reference's autograd (unet_stem2.npz)."""
    from ldm_image_generator_amd.ddpm import DDPM
    from ldm_image_generator_amd.train import L1LossFunction
    from ldm_image_generator_amd.unet import UNet
    from ldm_image_generator_amd import ops
    g = load_golden("unet_stem2")
    net = formula(UNet(input_channels=3, stages=[1, 2], channels=[32, 64], stem_size=2)).train()
    d = DDPM(model=net)
    x, t, e = T(g["loss_x"]).cuda(), T(g["loss_t"]), T(g["loss_e"]).cuda()
    ab = d.alpha_bar[t]
    xt = torch.empty_like(x)
    ops.qsample(x, e, torch.sqrt(ab).cuda(), torch.sqrt(1 - ab).cuda(), xt)
    random.seed(5)
    loss = L1LossFunction.apply(net(x=xt, time=t.cuda(), condition=None), e)
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    norms = dict(zip([str(n) for n in g["grad_names"]], g["grad_norms"]))
    for k, p in net.named_parameters():
        ref = norms[k]
        if ref < 0:
            assert p.grad is None, k
            continue
        assert p.grad is not None, k
        got = float(p.grad.double().norm())
        assert abs(got - ref) / max(ref, 1e-12) < 2e-4, (k, got, ref)
    assert rel_l2(net.encoder_first.weight.grad.cpu(), T(g["grad_encoder_first_weight"])) < 1e-4
    assert rel_l2(net.decoder_last.weight.grad.cpu(), T(g["grad_decoder_last_weight"])) < 1e-4
    assert rel_l2(net.decoder_last.bias.grad.cpu(), T(g["grad_decoder_last_bias"])) < 1e-4


def test_calculate_loss_end_to_end_and_optimizer_step(gpu_device):
    """ddpm.calculate_loss(x).backward() + AdamW step (train_ldm.py:67,81-86): loss finite, decreases on a fixed batch."""
    from ldm_image_generator_amd.ddpm import DDPM
    from ldm_image_generator_amd.unet import UNet
    net = formula(UNet(input_channels=3, stages=[1, 2], channels=[32, 64])).train()
    d = DDPM(model=net)
    opt = torch.optim.AdamW(d.parameters(), lr=2e-3)
    x = (torch.rand(16, 3, 32, 32, generator=torch.Generator().manual_seed(0)) * 2 - 1).cuda()     # BASELINE cfg 1 shape
    losses = []
    for it in range(6):
        random.seed(100)
        torch.manual_seed(100)                      # same t, e, experts every iteration: a fixed objective
        opt.zero_grad()
        loss = d.calculate_loss(x)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_one_epoch_of_train_ddpm_on_its_own_shape(gpu_device, precision):
    """BASELINE configs[0] on the HIP path: train_ddpm.py's loop (train_ddpm.py:11-16,28,31-45) -- default-width
    UNet(input_channels=3) inside DDPM, 64 synthetic 32x32 RGB images, batch 16, RAdam lr 1e-4, one epoch = 4 iterations.
    32x32 pixels put the deepest stage at 4x4 (attention windows clipped to the image); every iteration must give a finite
    loss and a finite update of every parameter that received a gradient."""
    from ldm_image_generator_amd import train
    from ldm_image_generator_amd.ddpm import DDPM
    from ldm_image_generator_amd.unet import UNet
    net = formula(UNet(input_channels=3)).train()
    train.set_precision(net, precision)
    d = DDPM(model=net)
    opt = torch.optim.RAdam(d.parameters(), lr=1e-4)
    images = (torch.rand(64, 3, 32, 32, generator=torch.Generator().manual_seed(0)) * 2 - 1)
    before = {k: p.detach().clone() for k, p in net.named_parameters()}
    random.seed(1)
    torch.manual_seed(1)
    losses = []
    for it in range(4):
        opt.zero_grad()
        loss = d.calculate_loss(images[16 * it:16 * it + 16].cuda())
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and all(0.0 < v < 10.0 for v in losses), losses
    moved = 0
    for k, p in net.named_parameters():
        assert bool(torch.isfinite(p).all()), k
        moved += int(not torch.equal(p.detach(), before[k]))
    assert moved > 500


def test_unet_gradients_vs_oracle_autograd(gpu_device):
    """Fresh inputs, per-sample timesteps, eval mode (all blocks): every parameter gradient vs oracle autograd."""
    from ldm_image_generator_amd import synth
    from ldm_image_generator_amd.unet import UNet
    cfg = dict(input_channels=8, stages=[1, 2], channels=[32, 64])
    net = formula(UNet(**cfg)).eval()
    sd = {k: v.clone().requires_grad_() for k, v in synth.fill_state_dict(net.state_dict()).items()}
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(4, 8, 16, 16, generator=gen)
    t = torch.tensor([5, 900, 5, 333])
    dy = torch.randn(4, 8, 16, 16, generator=gen)
    random.seed(8)
    O.unet_forward(sd, x, t, stages=cfg["stages"], channels=cfg["channels"], training=False).backward(dy)
    random.seed(8)
    net(x.cuda(), t.cuda()).backward(dy.cuda())
    for k, p in net.named_parameters():
        ref = sd[k].grad
        if ref is None or float(ref.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert rel_l2(p.grad.cpu(), ref) < 1e-4, k


# ------------------------------------------------------------------------------------------------------
# FULL width (BASELINE cfg 5: default UNet(), C = 128..1024): the routes the tiny nets never take -- TN weight-gradient
# kernel with fused column sums and its split choice, the grouped-conv weight-gradient kernel, 121-window attention
# backward at R = 64, one FiLM slot per sample -- against the reference's own autograd (tests/golden/loss_full.npz).
# ------------------------------------------------------------------------------------------------------
def _full_step(net, d, g, tag, seed):
    from ldm_image_generator_amd import ops
    from ldm_image_generator_amd.train import L1LossFunction
    x, t, e = T(g["x_" + tag]).cuda(), T(g["t_" + tag]), T(g["e_" + tag]).cuda()
    ab = d.alpha_bar[t]
    xt = torch.empty_like(x)
    ops.qsample(x, e, torch.sqrt(ab).cuda(), torch.sqrt(1 - ab).cuda(), xt)
    for p in net.parameters():
        p.grad = None
    random.seed(seed)
    e_theta = net(x=xt, time=t.cuda(), condition=None)
    loss = L1LossFunction.apply(e_theta, e)
    loss.backward()
    return loss


@pytest.fixture(scope="module")
def full_net(gpu_device):
    from ldm_image_generator_amd.unet import UNet
    return formula(UNet()).train()


@pytest.mark.parametrize("tag,seed", [("r64", 5), ("r32", 6)])
def test_full_width_training_step_matches_reference_gradients(gpu_device, full_net, tag, seed):
    from ldm_image_generator_amd.ddpm import DDPM
    g = load_golden("loss_full")
    net = full_net
    d = DDPM(model=net)
    loss = _full_step(net, d, g, tag, seed)
    ref_loss = float(g["loss_" + tag])
    assert abs(float(loss.detach()) - ref_loss) < 1e-5 * abs(ref_loss)
    names = [str(n) for n in g["grad_names"]]
    norms = dict(zip(names, g["grad_norms_" + tag]))
    worst, n_used = 0.0, 0
    for k, p in net.named_parameters():
        ref = norms[k]
        if ref < 0:
            assert p.grad is None, k
            continue
        assert p.grad is not None, k
        n_used += 1
        got = float(p.grad.double().norm())
        err = abs(got - ref) / max(ref, 1e-12)
        worst = max(worst, err)
        assert err < 2e-4, (k, got, ref)
    assert n_used > 500
    print("full-width %s: %d used parameters, worst gradient-norm deviation %.2e" % (tag, n_used, worst))
    grads = dict(net.named_parameters())
    assert rel_l2(net.encoder_first.weight.grad.cpu(), T(g["grad_encoder_first_weight_" + tag])) < 1e-4
    assert rel_l2(net.decoder_last.weight.grad.cpu(), T(g["grad_decoder_last_weight_" + tag])) < 1e-4
    checked = 0
    for key in g:
        if key.startswith("gslice_") and key.endswith("_" + tag) and not key.endswith("_key"):
            pname = str(g[key + "_key"])
            gk = grads[pname].grad
            sl = gk.reshape(gk.shape[0], -1)[:64, :96] if gk.ndim > 1 else gk[:256]
            assert rel_l2(sl.cpu(), T(g[key])) < 1e-4, (key, pname)
            checked += 1
    assert checked >= 6


def test_full_width_batch_step_equals_mean_of_half_batches(gpu_device, full_net):
    """Size-independent property of the training step (mean-reduced L1 loss, no batch statistics anywhere): the
    gradients of a batch are the average of the gradients of its two halves under the same expert / depth decisions."""
    from ldm_image_generator_amd import ops
    from ldm_image_generator_amd.train import L1LossFunction
    net = full_net
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(8, 8, 32, 32, generator=gen).cuda()
    e = torch.randn(8, 8, 32, 32, generator=gen).cuda()
    t = torch.randint(1, 1000, (8,), generator=gen).cuda()

    def run(sl):
        for p in net.parameters():
            p.grad = None
        random.seed(77)
        loss = L1LossFunction.apply(net(x=x[sl], time=t[sl], condition=None), e[sl])
        loss.backward()
        return float(loss), {k: (None if p.grad is None else p.grad.clone()) for k, p in net.named_parameters()}

    l_all, g_all = run(slice(0, 8))
    l_a, g_a = run(slice(0, 4))
    l_b, g_b = run(slice(4, 8))
    assert abs(l_all - 0.5 * (l_a + l_b)) < 1e-5 * abs(l_all)
    worst = 0.0
    for k, ga in g_all.items():
        if ga is None:
            assert g_a[k] is None and g_b[k] is None, k
            continue
        err = rel_l2(0.5 * (g_a[k] + g_b[k]), ga)
        worst = max(worst, err)
        assert err < 2e-4, (k, err)
    print("half-batch property: worst rel-L2 %.2e" % worst)
