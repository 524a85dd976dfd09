"""Parity of each HIP entry point (through the C ABI) against a CPU statement of the
same op.  Tolerances (fp32 MFMA == fmaf chain; reference self-noise ~5e-7):
GEMM-family kernels rel-L2 <= 1e-5 against an fp64 evaluation, elementwise kernels
bit-exact where the reference's op order is reproduced exactly."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import T, load_golden, rel_l2
from oracle import ldm_oracle as O

pytestmark = pytest.mark.gpu

KTOL = 1e-5


@pytest.fixture(scope="module", params=["stream", "tile"])
def ops(gpu_device, request):
    """Every kernel test runs under both GEMM schedules (persistent LDS-DMA stream / one tile per workgroup)."""
    from ldm_image_generator_amd import ops as _ops
    from ldm_image_generator_amd import _lib
    assert _lib.load().ldm_device_ok() == 1, "device 0 is not gfx950"
    old = _ops.gemm_variant(1 if request.param == "stream" else 0)
    yield _ops
    _ops.gemm_variant(old)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).float()


@pytest.mark.parametrize("M,N,K", [(1, 32, 32), (37, 64, 96), (128, 128, 64), (300, 256, 512), (16, 384, 2048),
                                   (1000, 96, 160), (129, 128, 32)])
@pytest.mark.parametrize("act", ["none", "relu"])
def test_gemm_plain(ops, gpu_device, M, N, K, act):
    a, w, b, add = rnd(M, K), rnd(N, K, seed=1, scale=K ** -0.5), rnd(N, seed=2), rnd(M, N, seed=3)
    out = torch.empty(M, N, device=gpu_device)
    ops.gemm(a.cuda(), M, N, K, [w.cuda()], out, biases=[b.cuda()], addend=add.cuda(),
             act=ops.ACT_RELU if act == "relu" else ops.ACT_NONE)
    ref = a.double() @ w.double().t() + b.double()
    if act == "relu":
        ref = torch.relu(ref)
    ref = ref + add.double()
    assert rel_l2(out.cpu(), ref) < KTOL


@pytest.mark.parametrize("M,C", [(50, 32), (260, 64), (257, 128), (64, 256)])
def test_gemm_gate_and_ksegments(ops, gpu_device, M, C):
    """ReGLU sum of 3 experts: gated GEMM with N-segments, then GEMM with K-segments (modules.py:15,36)."""
    x = rnd(M, C)
    wa = [rnd(C, C, seed=10 + i, scale=C ** -0.5) for i in range(3)]
    wb = [rnd(C, C, seed=20 + i, scale=C ** -0.5) for i in range(3)]
    wc = [rnd(C, C, seed=30 + i, scale=C ** -0.5) for i in range(3)]
    ba = [rnd(C, seed=40 + i) for i in range(3)]
    bb = [rnd(C, seed=50 + i) for i in range(3)]
    bc = [rnd(C, seed=60 + i) for i in range(3)]
    res = rnd(M, C, seed=70)
    dev = lambda ts: [t.cuda() for t in ts]
    hid = torch.empty(M, 3 * C, device=gpu_device)
    ops.gemm(x.cuda(), M, 3 * C, C, dev(wa), hid, weights2=dev(wb), biases=dev(ba), biases2=dev(bb), act=ops.ACT_GATE)
    xd = x.double()
    href = torch.cat([(xd @ wa[i].double().t() + ba[i].double()) * torch.relu(xd @ wb[i].double().t() + bb[i].double())
                      for i in range(3)], dim=1)
    assert rel_l2(hid.cpu(), href) < KTOL
    out = res.cuda().clone()
    ops.gemm(hid, M, C, 3 * C, dev(wc), out, biases=dev(bc), seg_mode=ops.SEG_K, addend=out)
    ref = sum(href[:, i * C:(i + 1) * C] @ wc[i].double().t() + bc[i].double() for i in range(3)) + res.double()
    assert rel_l2(out.cpu(), ref) < KTOL


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 5, 7, 32, 32), (1, 16, 16, 64, 128), (3, 8, 8, 96, 64), (1, 33, 9, 32, 96)])
def test_conv3x3_dense(ops, gpu_device, B, H, W, Cin, Cout):
    x = rnd(B, Cin, H, W)
    w = rnd(Cout, Cin, 3, 3, seed=1, scale=(9 * Cin) ** -0.5)
    b = rnd(Cout, seed=2)
    res = rnd(B, Cout, H, W, seed=3)
    rows = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().cuda()
    add = res.permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().cuda()
    wp = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().cuda()
    out = torch.empty(B * H * W, Cout, device=gpu_device)
    ops.gemm(rows, B * H * W, Cout, 9 * Cin, [wp], out, lda=Cin, ldw=9 * Cin, biases=[b.cuda()], act=ops.ACT_LRELU,
             slope=0.01, addend=add, a_mode=ops.A_CONV3X3, conv_hw=(H, W), cin=Cin)
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), padding=1), 0.01) + res.double()
    got = out.cpu().reshape(B, H, W, Cout).permute(0, 3, 1, 2)
    assert rel_l2(got, ref) < KTOL


@pytest.mark.parametrize("B,H,W,C", [(2, 6, 9, 64), (1, 32, 32, 128), (2, 4, 4, 256), (1, 7, 5, 32)])
def test_grouped_conv3x3(ops, gpu_device, B, H, W, C):
    x = rnd(B, C, H, W)
    w = rnd(C, 32, 3, 3, seed=1, scale=288 ** -0.5)
    b = rnd(C, seed=2)
    res = rnd(B, C, H, W, seed=3)
    rows = x.permute(0, 2, 3, 1).reshape(-1, C).contiguous().cuda()
    add = res.permute(0, 2, 3, 1).reshape(-1, C).contiguous().cuda()
    wp = w.permute(0, 2, 3, 1).reshape(C, 288).contiguous().cuda()
    out = torch.empty(B * H * W, C, device=gpu_device)
    ops.gemm(rows, B * H * W, 32, 288, [wp], out, lda=C, ldw=288, biases=[b.cuda()], addend=add, ldadd=C, ldo=C,
             a_mode=ops.A_CONV3X3, conv_hw=(H, W), cin=32, groups=C // 32, a_gstride=32, w_gstride=32 * 288, o_gstride=32, b_gstride=32)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1, groups=C // 32) + res.double()
    got = out.cpu().reshape(B, H, W, C).permute(0, 3, 1, 2)
    assert rel_l2(got, ref) < KTOL


def test_grouped_conv3x3_dedicated_kernel_bit_identical(gpu_device):
    """The stream schedule routes 32-per-group 3x3 convolutions to their own kernel (gconv.hip: pixel block + weights
    resident in LDS, borders as zero-row reads); the tile schedule keeps the generic implicit-im2col GEMM.  Same
    k-order -> bitwise equal, on ragged tiles, several images per tile, wide and tiny maps, with and without addend."""
    from ldm_image_generator_amd import ops as o
    for (B, H, W, C, act, with_add) in [(3, 32, 32, 128, o.ACT_NONE, True), (5, 7, 5, 64, o.ACT_RELU, True),
                                        (70, 4, 4, 256, o.ACT_NONE, False), (2, 12, 64, 64, o.ACT_LRELU, True),
                                        (64, 4, 4, 1024, o.ACT_NONE, True), (16, 16, 16, 512, o.ACT_RELU, False),
                                        (1, 40, 96, 64, o.ACT_NONE, True)]:
        M = B * H * W
        rows = rnd(M, C).cuda()
        add = rnd(M, C, seed=3).cuda() if with_add else None
        wp = rnd(C, 288, seed=1, scale=288 ** -0.5).cuda()
        b = rnd(C, seed=2).cuda()
        outs = []
        for v in (0, 1):
            old = o.gemm_variant(v)
            out = torch.zeros(M, C, device=gpu_device)
            o.gemm(rows, M, 32, 288, [wp], out, lda=C, ldw=288, biases=[b], addend=add, ldadd=C, ldo=C, act=act, slope=0.01,
                   a_mode=o.A_CONV3X3, conv_hw=(H, W), cin=32, groups=C // 32, a_gstride=32, w_gstride=32 * 288, o_gstride=32, b_gstride=32)
            outs.append(out)
            o.gemm_variant(old)
        assert torch.equal(outs[0], outs[1]), (B, H, W, C)
        xn = rows.cpu().double().reshape(B, H, W, C).permute(0, 3, 1, 2)
        ref = F.conv2d(xn, wp.cpu().double().reshape(C, 3, 3, 32).permute(0, 3, 1, 2), b.cpu().double(), padding=1, groups=C // 32)
        ref = F.relu(ref) if act == o.ACT_RELU else (F.leaky_relu(ref, 0.01) if act == o.ACT_LRELU else ref)
        ref = ref.permute(0, 2, 3, 1).reshape(M, C) + (add.cpu().double() if with_add else 0)
        assert rel_l2(outs[1].cpu().double(), ref) < KTOL, (B, H, W, C)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 3, 5, 64, 32), (1, 8, 8, 128, 64)])
def test_conv_transpose_2x2(ops, gpu_device, B, H, W, Cin, Cout):
    x = rnd(B, Cin, H, W)
    w = rnd(Cin, Cout, 2, 2, seed=1, scale=Cin ** -0.5)
    b = rnd(Cout, seed=2)
    rows = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().cuda()
    wp = w.permute(2, 3, 1, 0).reshape(4 * Cout, Cin).contiguous().cuda()
    out = torch.empty(B * 4 * H * W, Cout, device=gpu_device)
    ops.gemm(rows, B * H * W, 4 * Cout, Cin, [wp], out, biases=[b.cuda()], ldo=Cout, o_mode=ops.O_CONVT2X2,
             out_hw=(H, W), cout=Cout)
    ref = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2)
    got = out.cpu().reshape(B, 2 * H, 2 * W, Cout).permute(0, 3, 1, 2)
    assert rel_l2(got, ref) < KTOL


def test_upsample_conv_skip(ops, gpu_device):
    """unet.py:85,101: Conv1x1(nearest_up2(x)) + skip, computed at the coarse grid and replicated."""
    B, H, W, Cin, Cout = 2, 4, 3, 64, 32
    x, w, b, skip = rnd(B, Cin, H, W), rnd(Cout, Cin, seed=1, scale=Cin ** -0.5), rnd(Cout, seed=2), rnd(B, Cout, 2 * H, 2 * W, seed=3)
    rows = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().cuda()
    srows = skip.permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().cuda()
    out = torch.empty(B * 4 * H * W, Cout, device=gpu_device)
    ops.gemm(rows, B * H * W, Cout, Cin, [w.cuda()], out, biases=[b.cuda()], addend=srows, o_mode=ops.O_UP2, out_hw=(H, W))
    up = F.interpolate(x.double(), scale_factor=2, mode="nearest")
    ref = F.conv2d(up, w.double().reshape(Cout, Cin, 1, 1), b.double()) + skip.double()
    got = out.cpu().reshape(B, 2 * H, 2 * W, Cout).permute(0, 3, 1, 2)
    assert rel_l2(got, ref) < KTOL


@pytest.mark.parametrize("mode,B,H,W,Cin,Cout,with_add,act", [
    ("up2", 2, 4, 4, 128, 64, True, 0),          # UNet decoder s3 -> s2 shape class: OW = 4 < rows per store instruction
    ("up2", 3, 5, 7, 64, 128, True, 0),          # ragged M (105 rows), 64-column strips
    ("up2", 8, 32, 32, 256, 128, True, 0),       # 128x128 tiles (>= 512 of them)
    ("up2", 1, 6, 10, 64, 64, False, 1),
    ("convt", 2, 3, 5, 64, 64, False, 0),        # N = 256, Cout = 64: a 128-column tile spans two quadrants
    ("convt", 1, 8, 8, 128, 32, True, 3),        # Cout = 32: four quadrants per 128 columns
    ("convt", 16, 32, 32, 128, 128, False, 0),   # 128x128 tiles
])
def test_scatter_outputs_wide_epilogue_bit_identical_to_direct(ops, gpu_device, mode, B, H, W, Cin, Cout, with_add, act):
    """up x2 / convT 2x2 outputs through the 16-byte scatter epilogue == the direct 4-byte one (same arithmetic, same order), and both
    against fp64."""
    o = ops
    M = B * H * W
    x = rnd(B, Cin, H, W)
    rows = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().cuda()
    add = rnd(B * 4 * H * W, Cout, seed=3).cuda() if with_add else None
    if mode == "up2":
        w, b = rnd(Cout, Cin, seed=1, scale=Cin ** -0.5), rnd(Cout, seed=2)
        wp, N, kw = w.cuda(), Cout, dict(o_mode=o.O_UP2, out_hw=(H, W))
        ref = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), w.double().reshape(Cout, Cin, 1, 1), b.double())
    else:
        w, b = rnd(Cin, Cout, 2, 2, seed=1, scale=Cin ** -0.5), rnd(Cout, seed=2)
        wp, N = w.permute(2, 3, 1, 0).reshape(4 * Cout, Cin).contiguous().cuda(), 4 * Cout
        kw = dict(ldo=Cout, ldadd=Cout, o_mode=o.O_CONVT2X2, out_hw=(H, W), cout=Cout)
        ref = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2)
    ref = F.relu(ref) if act == o.ACT_RELU else (F.leaky_relu(ref, 0.1) if act == o.ACT_LRELU else ref)
    ref = ref.permute(0, 2, 3, 1).reshape(-1, Cout) + (add.cpu().double() if with_add else 0)
    outs = []
    for wide in (0, 1):
        old = o.gemm_wide_epilogue(wide)
        try:
            out = torch.full((B * 4 * H * W, Cout), float("nan"), device=gpu_device)
            o.gemm(rows, M, N, Cin, [wp], out, biases=[b.cuda()], addend=add, act=act, slope=0.1, **kw)
        finally:
            o.gemm_wide_epilogue(old)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    assert rel_l2(outs[1].cpu().double(), ref) < KTOL


@pytest.mark.parametrize("B,HW,C", [(2, 35, 32), (3, 16, 96), (2, 64, 128), (1, 9, 512), (2, 4, 1024)])
def test_channelnorm_film(ops, gpu_device, B, HW, C):
    x = rnd(B * HW, C) * 3 + 0.5
    film = rnd(2 * HW, 2 * C, seed=1)
    slot = torch.tensor([1, 0, 1][:B], dtype=torch.int32)
    out = torch.empty(B * HW, C, device=gpu_device)
    ops.channelnorm_film(x.cuda(), film.cuda(), slot.cuda(), out, B, HW, C, 1e-4)
    xn = O.channel_norm(x.reshape(B, HW, C).permute(0, 2, 1).reshape(B, C, HW, 1)).reshape(B, C, HW).permute(0, 2, 1)
    f = film.reshape(2, HW, 2 * C)[slot.long()]
    ref = xn * f[:, :, :C] + f[:, :, C:]
    assert rel_l2(out.cpu().reshape(B, HW, C), ref) < 2e-6


def test_sincos_embed_matches_reference_tables(gpu_device):
    from ldm_image_generator_amd import sinusoidal
    g = load_golden("tables")
    steps = T(g["te_steps"]).long()
    for c, h, w in [(32, 7, 5), (128, 32, 32), (1024, 4, 4)]:
        emb = sinusoidal.embed(steps.cuda(), h, w, c).cpu().reshape(50, h * w, 2 * c)
        pe = T(g["pe_%d_%d_%d" % (c, h, w)]).permute(1, 2, 0).reshape(h * w, c)
        assert (emb[0, :, :c] - pe).abs().max() < 2.4e-7
        assert torch.equal(emb[0, :, :c], emb[49, :, :c])
        if "te_%d" % c in g:
            te = T(g["te_%d" % c])
            # the argument (float(t) * pi_f32) * f is formed in the reference's order from host-computed frequencies, so it is
            # bit-identical; device sinf/cosf and torch's CPU sin/cos then differ by at most one ulp of the RESULT
            # (measured on MI355X, tools/sincos_err.py: max 5.96e-8, 81 % of the values bit-equal)
            assert (emb[:, 0, c:] - te).abs().max() < 2.4e-7
            assert (emb[:, 0, c:] - te).abs().mean() < 3e-8
            assert torch.equal(emb[:, 0, c:], emb[:, h * w - 1, c:])


def test_avgpool_stem_head_transposes(ops, gpu_device):
    B, H, W, C = 2, 6, 4, 64
    x = rnd(B, C, H, W)
    rows = torch.empty(B * H * W, C, device=gpu_device)
    ops.nchw_to_nhwc(x.cuda(), rows, B, C, H * W)
    assert torch.equal(rows.cpu(), x.permute(0, 2, 3, 1).reshape(-1, C))
    back = torch.empty(B, C, H, W, device=gpu_device)
    ops.nhwc_to_nchw(rows, back, B, C, H * W)
    assert torch.equal(back.cpu(), x)
    pooled = torch.empty(B * (H // 2) * (W // 2), C, device=gpu_device)
    ops.avgpool2(rows, pooled, B, H, W, C)
    ref = F.avg_pool2d(x, 2).permute(0, 2, 3, 1).reshape(-1, C)
    assert rel_l2(pooled.cpu(), ref) < 1e-6
    for cin in (3, 8):
        xi = rnd(B, cin, H, W, seed=4)
        w, b = rnd(C, cin, seed=5), rnd(C, seed=6)
        out = torch.empty(B * H * W, C, device=gpu_device)
        ops.stem_nchw(xi.cuda(), w.cuda(), b.cuda(), out, B, cin, H * W, C)
        ref = (torch.einsum("oc,nchw->nhwo", w.double(), xi.double()) + b.double()).reshape(-1, C)
        assert rel_l2(out.cpu(), ref) < 1e-6
        wt, bt = rnd(C, cin, seed=7), rnd(cin, seed=8)
        o2 = torch.empty(B, cin, H, W, device=gpu_device)
        ops.head_nchw(rows, wt.cuda(), bt.cuda(), o2, B, C, H * W, cin)
        ref = torch.einsum("co,nchw->nohw", wt.double(), x.double()) + bt.double().reshape(1, -1, 1, 1)
        assert rel_l2(o2.cpu(), ref) < 1e-6
    # C0 > 128 exercises the chunked head
    C2 = 320
    x2 = rnd(B, C2, H, W, seed=9)
    r2 = x2.permute(0, 2, 3, 1).reshape(-1, C2).contiguous().cuda()
    wt, bt = rnd(C2, 8, seed=10), rnd(8, seed=11)
    o2 = torch.empty(B, 8, H, W, device=gpu_device)
    ops.head_nchw(r2, wt.cuda(), bt.cuda(), o2, B, C2, H * W, 8)
    ref = torch.einsum("co,nchw->nohw", wt.double(), x2.double()) + bt.double().reshape(1, -1, 1, 1)
    assert rel_l2(o2.cpu(), ref) < 1e-6


def test_ddim_update_bit_exact(ops, gpu_device):
    _, _, alpha = O.schedule_tables()
    x, e, nz = rnd(2, 8, 16, 16) * 50, rnd(2, 8, 16, 16, seed=1) * 7, rnd(2, 8, 16, 16, seed=2)
    for (t, tn, eta) in [(999, 978, 0.0), (20, 0, 0.0), (0, 0, 0.0), (489, 468, 0.5)]:
        sigma, s1, s2, s3, s4 = O.ddim_coefficients(alpha, t, tn, eta)
        x_t0 = (x - s1 * e) / s2
        ref = x_t0 if t == 0 else s3 * x_t0 + s4 * e + sigma * nz
        got = x.cuda().clone()
        ops.ddim_update(got, e.cuda(), nz.cuda(), float(s1), float(s2), float(s3), float(s4), float(sigma), t == 0)
        assert torch.equal(got.cpu(), ref), (t, tn, eta)


def test_qsample_bit_exact(ops, gpu_device):
    _, alpha_bar, _ = O.schedule_tables()
    x, e = rnd(3, 8, 8, 8), rnd(3, 8, 8, 8, seed=1)
    t = torch.tensor([1, 500, 999])
    ab = alpha_bar[t].reshape(-1, 1, 1, 1)
    ref = torch.sqrt(ab) * x + torch.sqrt(1 - ab) * e
    out = torch.empty(3, 8, 8, 8, device=gpu_device)
    ops.qsample(x.cuda(), e.cuda(), torch.sqrt(alpha_bar[t]).cuda(), torch.sqrt(1 - alpha_bar[t]).cuda(), out)
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("C", [32, 64, 512])
def test_rgb_head_and_bilinear(ops, gpu_device, C):
    B, H, W = 2, 6, 10
    x = rnd(B, C, H, W)
    w, b = rnd(3, C, seed=1, scale=C ** -0.5), rnd(3, seed=2)
    prev = rnd(B, 3, H // 2, W // 2, seed=3)
    rows = x.permute(0, 2, 3, 1).reshape(-1, C).contiguous().cuda()
    out = torch.empty(B, 3, H, W, device=gpu_device)
    ops.rgb_head(rows, w.cuda(), b.cuda(), None, out, B, H, W, C)
    rgb = F.conv2d(x.double(), w.double().reshape(3, C, 1, 1), b.double())
    assert rel_l2(out.cpu(), rgb) < 2e-6
    ops.rgb_head(rows, w.cuda(), b.cuda(), prev.cuda(), out, B, H, W, C)
    ref = F.interpolate(prev.double(), scale_factor=2, mode="bilinear") + rgb
    assert rel_l2(out.cpu(), ref) < 2e-6


def test_to_uint8(ops, gpu_device):
    img = rnd(2, 3, 9, 7) * 0.8
    img[0, 0, 0, :4] = torch.tensor([-1.0, 1.0, 3.0, -2.0])
    out = torch.empty(2, 9, 7, 3, dtype=torch.uint8, device=gpu_device)
    ops.to_uint8_hwc(img.cuda(), out, 2, 3, 63)
    assert np.array_equal(out.cpu().numpy(), O.to_uint8_hwc(img))


def test_errors_are_reported_not_thrown(ops, gpu_device):
    from ldm_image_generator_amd._lib import LdmHipError, LdmHipUnavailable
    a = torch.zeros(4, 40, device=gpu_device)
    with pytest.raises(LdmHipError):
        ops.gemm(a, 4, 32, 40, [torch.zeros(32, 40, device=gpu_device)], torch.zeros(4, 32, device=gpu_device))   # K % 32
    with pytest.raises(LdmHipUnavailable):
        ops.avgpool2(torch.zeros(1, 4, 4, 8), torch.zeros(1, 2, 2, 8), 1, 4, 4, 8)                                   # CPU tensor


def test_gemm_pointer_table_groups(ops, gpu_device):
    """Independent layers batched on grid.y through device pointer tables (UNet._films)."""
    G, M, N, K = 5, 48, 128, 64
    a = rnd(M, K)
    ws = [rnd(N, K, seed=100 + g, scale=K ** -0.5).cuda() for g in range(G)]
    bs = [rnd(N, seed=200 + g).cuda() for g in range(G)]
    wt, bt = ops.pointer_table(ws), ops.pointer_table(bs)
    hid = torch.empty(G, M, N, device=gpu_device)
    ops.gemm(a.cuda(), M, N, K, None, hid, w_table=wt, bias_table=bt, act=ops.ACT_RELU, groups=G, a_gstride=0, o_gstride=M * N)
    for g in range(G):
        ref = torch.relu(a.double() @ ws[g].cpu().double().t() + bs[g].cpu().double())
        assert rel_l2(hid[g].cpu(), ref) < KTOL
    # second layer: per-group A
    w2 = [rnd(64, N, seed=300 + g, scale=N ** -0.5).cuda() for g in range(G)]
    wt2 = ops.pointer_table(w2)
    out = torch.empty(G, M, 64, device=gpu_device)
    ops.gemm(hid, M, 64, N, None, out, w_table=wt2, groups=G, a_gstride=M * N, o_gstride=M * 64)
    for g in range(G):
        assert rel_l2(out[g].cpu(), hid[g].cpu().double() @ w2[g].cpu().double().t()) < KTOL


def test_gemm_schedules_bit_identical(gpu_device):
    """The stream and tile-per-workgroup schedules run the same fmaf chains: outputs must match bitwise,
    including many-tiles-per-workgroup problems (persistent loop, ragged M, short K)."""
    from ldm_image_generator_amd import ops as o
    for (M, N, K) in [(70000, 128, 128), (33000, 384, 32), (4100, 1024, 64), (257, 2048, 96)]:
        a, w, b = rnd(M, K).cuda(), rnd(N, K, seed=1, scale=K ** -0.5).cuda(), rnd(N, seed=2).cuda()
        add = rnd(M, N, seed=3).cuda()
        outs = []
        for v in (0, 1):
            old = o.gemm_variant(v)
            out = torch.empty(M, N, device=gpu_device)
            o.gemm(a, M, N, K, [w], out, biases=[b], addend=add)
            outs.append(out)
            o.gemm_variant(old)
        assert torch.equal(outs[0], outs[1]), (M, N, K)
        ref = a.double() @ w.double().t() + b.double() + add.double()
        assert rel_l2(outs[1].double(), ref) < KTOL


def test_gemm_split_schedule_fp32_level_error(gpu_device):
    """Schedule 2 (fp32 operands cut exactly into three bf16 pieces, six bf16 MFMAs per product, fp32 accumulate):
    same stated kernel tolerance as the exact-fp32 schedules, and its error against fp64 stays within a small
    multiple of theirs.  Covers plain / K-segment / gated / 3x3-conv instances, ragged M, many tiles per workgroup."""
    from ldm_image_generator_amd import ops as o

    def both(fn):
        outs = []
        for v in (1, 2):
            old = o.gemm_variant(v)
            outs.append(fn())
            o.gemm_variant(old)
        return outs

    def check(outs, ref, what):
        e1, e2 = rel_l2(outs[0].double().cpu(), ref), rel_l2(outs[1].double().cpu(), ref)
        assert e2 < KTOL and e2 < 4 * e1 + 1e-7, (what, e1, e2)
        assert not torch.equal(outs[0], outs[1]), what + ": schedule 2 did not run"

    for (M, N, K) in [(70000, 128, 128), (4100, 1024, 64), (257, 256, 96), (33000, 384, 32), (9000, 64, 160)]:
        a, w, b = rnd(M, K).cuda(), rnd(N, K, seed=1, scale=K ** -0.5).cuda(), rnd(N, seed=2).cuda()
        add = rnd(M, N, seed=3).cuda()

        def plain():
            out = torch.empty(M, N, device=gpu_device)
            o.gemm(a, M, N, K, [w], out, biases=[b], addend=add)
            return out
        check(both(plain), (a.double() @ w.double().t() + b.double() + add.double()).cpu(), ("plain", M, N, K))
    # gated, N-segments by pointer + K-segment second GEMM (the RandomMoE pair)
    M, C = 5000, 128
    x = rnd(M, C).cuda()
    wa = [rnd(C, C, seed=10 + i, scale=C ** -0.5).cuda() for i in range(3)]
    wb = [rnd(C, C, seed=20 + i, scale=C ** -0.5).cuda() for i in range(3)]
    wc = [rnd(C, C, seed=30 + i, scale=C ** -0.5).cuda() for i in range(3)]
    ba = [rnd(C, seed=40 + i).cuda() for i in range(3)]

    def gate():
        hid = torch.empty(M, 3 * C, device=gpu_device)
        o.gemm(x, M, 3 * C, C, wa, hid, weights2=wb, biases=ba, biases2=ba, act=o.ACT_GATE)
        return hid
    hs = both(gate)
    xd = x.double()
    href = torch.cat([(xd @ wa[i].double().t() + ba[i].double()) * torch.relu(xd @ wb[i].double().t() + ba[i].double()) for i in range(3)], 1)
    check(hs, href.cpu(), "gate")

    def kseg():
        y = torch.empty(M, C, device=gpu_device)
        o.gemm(hs[0], M, C, 3 * C, wc, y, biases=ba, seg_mode=o.SEG_K, addend=x)
        return y
    yref = sum(hs[0][:, i * C:(i + 1) * C].double() @ wc[i].double().t() + ba[i].double() for i in range(3)) + xd
    check(both(kseg), yref.cpu(), "kseg")
    # dense 3x3 conv (VAE ResBlock shape class), implicit im2col
    B, R, C = 3, 20, 128
    M = B * R * R
    xc = rnd(M, C).cuda()
    w4 = rnd(C, C, 3, 3, seed=5, scale=(9 * C) ** -0.5)
    wk = w4.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous().cuda()
    bc = rnd(C, seed=6).cuda()

    def conv():
        y = torch.empty(M, C, device=gpu_device)
        o.gemm(xc, M, C, 9 * C, [wk], y, lda=C, ldw=9 * C, biases=[bc], act=o.ACT_LRELU, slope=0.01, addend=xc,
               a_mode=o.A_CONV3X3, conv_hw=(R, R), cin=C)
        return y
    xn = xc.cpu().double().reshape(B, R, R, C).permute(0, 3, 1, 2)
    cref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(xn, w4.double(), bc.cpu().double(), padding=1), 0.01)
    cref = cref.permute(0, 2, 3, 1).reshape(M, C) + xc.cpu().double()
    check(both(conv), cref, "conv3x3")
    # N = 64 dense conv (VAE stage-3 shape class)
    C2 = 64
    xc2 = rnd(M, C2).cuda()
    w42 = rnd(C2, C2, 3, 3, seed=7, scale=(9 * C2) ** -0.5)
    wk2 = w42.permute(0, 2, 3, 1).reshape(C2, 9 * C2).contiguous().cuda()

    def conv64():
        y = torch.empty(M, C2, device=gpu_device)
        o.gemm(xc2, M, C2, 9 * C2, [wk2], y, lda=C2, ldw=9 * C2, a_mode=o.A_CONV3X3, conv_hw=(R, R), cin=C2)
        return y
    xn2 = xc2.cpu().double().reshape(B, R, R, C2).permute(0, 3, 1, 2)
    check(both(conv64), torch.nn.functional.conv2d(xn2, w42.double(), padding=1).permute(0, 2, 3, 1).reshape(M, C2), "conv3x3 N=64")


@pytest.mark.parametrize("M,N,K,mode", [(16, 1024, 3072, "kseg"), (64, 512, 1536, "kseg"), (64, 1536, 512, "gate"), (16, 3072, 1024, "gate"),
                                         (100, 256, 1024, "plain"), (128, 4096, 2048, "relu")])
def test_gemm_split_k_small_m(ops, gpu_device, M, N, K, mode):
    """Few tiles + long reduction: the split-K path (partials in the caller's scratch, fixed-order sum) against fp64."""
    a = rnd(M, K)
    add = rnd(M, N, seed=5)
    if mode == "kseg":
        c = K // 3
        ws = [rnd(N, c, seed=10 + i, scale=K ** -0.5) for i in range(3)]
        bs = [rnd(N, seed=20 + i) for i in range(3)]
        out = add.cuda().clone()
        ops.gemm(a.cuda(), M, N, K, [w.cuda() for w in ws], out, biases=[b.cuda() for b in bs], seg_mode=ops.SEG_K, addend=out)
        ref = sum(a[:, i * c:(i + 1) * c].double() @ ws[i].double().t() + bs[i].double() for i in range(3)) + add.double()
    elif mode == "gate":
        f = N // 3
        wa = [rnd(f, K, seed=10 + i, scale=K ** -0.5) for i in range(3)]
        wb = [rnd(f, K, seed=20 + i, scale=K ** -0.5) for i in range(3)]
        ba = [rnd(f, seed=30 + i) for i in range(3)]
        bb = [rnd(f, seed=40 + i) for i in range(3)]
        out = torch.empty(M, N, device=gpu_device)
        dev = lambda ts: [t.cuda() for t in ts]
        ops.gemm(a.cuda(), M, N, K, dev(wa), out, weights2=dev(wb), biases=dev(ba), biases2=dev(bb), act=ops.ACT_GATE)
        ad = a.double()
        ref = torch.cat([(ad @ wa[i].double().t() + ba[i].double()) * torch.relu(ad @ wb[i].double().t() + bb[i].double()) for i in range(3)], 1)
    else:
        w, b = rnd(N, K, seed=1, scale=K ** -0.5), rnd(N, seed=2)
        out = torch.empty(M, N, device=gpu_device)
        ops.gemm(a.cuda(), M, N, K, [w.cuda()], out, biases=[b.cuda()], addend=add.cuda(), act=ops.ACT_RELU if mode == "relu" else ops.ACT_NONE)
        ref = a.double() @ w.double().t() + b.double()
        ref = (torch.relu(ref) if mode == "relu" else ref) + add.double()
    assert rel_l2(out.cpu(), ref) < KTOL


@pytest.mark.parametrize("M,N,K,mode", [
    (256, 256, 32, "plain"),            # one tile, two 16-k steps: ring never full
    (512, 128, 64, "plain128"),         # 256 x 128 tiles (one accumulator column per wave), four steps = exactly one ring
    (768, 384, 128, "nseg128"),         # three N-segments of 128 (QKV-like at C = 128): 256 x 128 tiles
    (1024, 768, 256, "nseg"),           # three N-segments of 256: 256 x 256 tiles, relu
    (768, 512, 1536, "kseg"),           # K-segments + bias sum + addend in place (the MoE's second GEMM)
    (2048, 384, 128, "gate"),           # ReGLU pair at C = 128: three experts by pointer, 128 hidden columns per tile
    (512, 1536, 512, "gate"),
    (65536, 256, 256, "lrelu"),         # 256 tiles on 256 workgroups ... and
    (131072, 128, 384, "kseg128"),      # ... 512 tiles of 256 x 128: the stream across tile boundaries, K-segments of 128
    (4096, 768, 96, "gate"),            # schedule 3 (eight workgroups): 12 gated tiles per workgroup of six steps each
    (8192, 384, 32, "nseg128"),         # ... 12 tiles of TWO steps (shorter than the ring)
    (4096, 512, 160, "plain"),          # ... ten steps per tile, 4 tiles per workgroup
    (6144, 256, 96, "kseg"),            # ... K-segments of 32 (two steps each), addend, 3 tiles per workgroup
])
def test_gemm_ring_kernel_fp32_bit_identical_to_stream_kernel(ops, gpu_device, M, N, K, mode):
    """One-workgroup-per-CU ring kernel (gemm_ring.hip, exact fp32) == the 128-row stream kernel bit for bit (same k pairing and
    order), and right against float64."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).cuda()
    kw, ref = {}, None
    ad = a.double()
    if mode in ("plain", "plain128", "lrelu"):
        w = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
        b = torch.randn(N, generator=g).cuda()
        ws, kw = [w], dict(biases=[b])
        ref = ad @ w.double().t() + b.double()
        if mode == "lrelu":
            kw.update(act=ops.ACT_LRELU, slope=0.2)
            ref = torch.where(ref > 0, ref, ref * 0.2)
    elif mode in ("nseg", "nseg128"):
        ws = [(torch.randn(N // 3, K, generator=g) * K ** -0.5).cuda() for _ in range(3)]
        bs = [torch.randn(N // 3, generator=g).cuda() for _ in range(3)]
        kw = dict(biases=bs, act=ops.ACT_RELU)
        ref = torch.relu(torch.cat([ad @ w.double().t() + b.double() for w, b in zip(ws, bs)], 1))
    elif mode in ("kseg", "kseg128"):
        c = K // 3
        ws = [(torch.randn(N, c, generator=g) * K ** -0.5).cuda() for _ in range(3)]
        bs = [torch.randn(N, generator=g).cuda() for _ in range(3)]
        kw = dict(biases=bs, seg_mode=ops.SEG_K)
        ref = sum(ad[:, i * c:(i + 1) * c] @ ws[i].double().t() + bs[i].double() for i in range(3))
    else:
        f = N // 3
        ws = [(torch.randn(f, K, generator=g) * K ** -0.5).cuda() for _ in range(3)]
        wb = [(torch.randn(f, K, generator=g) * K ** -0.5).cuda() for _ in range(3)]
        ba = [torch.randn(f, generator=g).cuda() for _ in range(3)]
        bb = [torch.randn(f, generator=g).cuda() for _ in range(3)]
        kw = dict(weights2=wb, biases=ba, biases2=bb, act=ops.ACT_GATE)
        ref = torch.cat([(ad @ ws[i].double().t() + ba[i].double()) * torch.relu(ad @ wb[i].double().t() + bb[i].double()) for i in range(3)], 1)
    base = torch.randn(M, N, generator=g).cuda() if mode.startswith("kseg") else None
    if base is not None:
        ref = ref + base.double()
    outs = {}
    old = ops.gemm_ring(1)
    try:
        for sched in (0, 2, 3):
            ops.gemm_ring(sched)
            out = base.clone() if base is not None else torch.full((M, N), float("nan"), device=gpu_device)
            ops.gemm(a, M, N, K, ws, out, addend=out if base is not None else None, **kw)
            outs[sched] = out
    finally:
        ops.gemm_ring(old)
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[0], outs[3])
    assert rel_l2(outs[2].double().cpu(), ref.cpu()) < KTOL


def test_gemm_split_schedule_on_nonfinite_and_denormal_operands(ops, gpu_device):
    """Schedule 2 (three exact bf16 pieces per fp32 value) against schedule 1 (exact fp32) on operands a checkpoint can contain:
    inf / NaN (non-finite results land in the SAME output elements; the split turns an inf operand into NaN because inf - inf appears
    in its residual), denormal operands (their low pieces are flushed: up to ~10 % RELATIVE error on results that are themselves
    ~1e-40 ... 1e-28, i.e. an absolute difference below 2e-29 here), values whose products overflow."""
    M, N, K = 256, 128, 128
    g = torch.Generator().manual_seed(9)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * K ** -0.5
    a[3, 5] = float("inf")
    a[7, 0] = float("nan")
    a[9, :] = 1e-40                                   # denormal row
    a[11, :] = 1e30
    w[4, :] = 1e10                                    # row 11 x column 4 overflows
    a[13, 17] = -float("inf")
    outs = {}
    old = ops.gemm_variant(1)
    try:
        for v in (1, 2):
            ops.gemm_variant(v)
            out = torch.empty(M, N, device=gpu_device)
            ops.gemm(a.cuda(), M, N, K, [w.cuda()], out)
            outs[v] = out.cpu()
    finally:
        ops.gemm_variant(old)
    f1, f2 = torch.isfinite(outs[1]), torch.isfinite(outs[2])
    assert torch.equal(f1, f2)                                             # same non-finite pattern ...
    assert not f1[3].any() and not f1[7].any() and not f1[13].any() and not bool(f1[11, 4])
    assert bool(f1[0].all()) and bool(f1[9].all())
    fin = f1.clone()
    fin[9] = False
    assert rel_l2(outs[2][fin], outs[1][fin]) < 2e-6                      # ... fp32-level agreement elsewhere
    assert float((outs[2][9] - outs[1][9]).abs().max()) < 2e-29           # denormal row (against weights up to 1e10)
