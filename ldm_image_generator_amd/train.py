"""Training step of the UNet on the HIP path: forward with saved activations + hand-written backward.

The reference trains through torch autograd (`ddpm.calculate_loss(x).backward()`, train_ldm.py:81-83).
Here the whole UNet is ONE ``torch.autograd.Function``: its forward runs the same kernels as inference
(with the ReGLU pre-activations kept), its backward walks a tape in reverse and produces the gradient of
every parameter that took part in the forward.  Parameters that did not (unselected experts, blocks
skipped by stochastic depth, the dead cross-attention) get ``None`` -- exactly what autograd gives the
reference, so AdamW skips them the same way.

Every GEMM-shaped gradient is an ``ldm_gemm_f32`` launch:
  data gradient    dX = dY . W        -> NT GEMM against the transposed weight (cached per weight version)
  weight gradient  dW = dY^T . X      -> NT GEMM over transposed activations, split along the huge reduction
                                         (pixels) into grid groups, partial sums reduced by a second kernel
"""
import weakref

import torch

from . import ops, weights


def _T(x, rows=None):
    """[M, C] -> [C, M] (device transpose kernel)."""
    m, c = x.shape
    out = torch.empty(c, m, device=x.device, dtype=torch.float32)
    ops.nhwc_to_nchw(x, out, 1, c, m)
    return out


def _T_colsum(x):
    """[M, C] -> ([C, M], column sums [C]) in one pass: the transposed activation for the weight-gradient GEMM and
    the bias gradient that goes with it."""
    m, c = x.shape
    out = torch.empty(c, m, device=x.device, dtype=torch.float32)
    csum = torch.empty(c, device=x.device, dtype=torch.float32)
    ops.transpose_colsum(x, out, csum)
    return out, csum


class _WeightT:
    """W [N, K] -> W^T [K, N], cached per parameter version."""

    def __init__(self):
        self.cache = {}

    def get(self, w2d, key_tensor):
        key = weights.key(key_tensor)
        hit = self.cache.get(id(key_tensor))
        # ids (and allocator addresses) are reused after a module dies: an entry is only valid for the very object it was made for,
        # and dies with it (weakref callback)
        if hit is None or hit[0] != key or hit[2]() is not key_tensor:
            kid = id(key_tensor)
            hit = (key, _T(w2d.contiguous()), weakref.ref(key_tensor, lambda _r, kid=kid, c=self.cache: c.pop(kid, None)))
            self.cache[kid] = hit
        return hit[1]


WT = _WeightT()


def _w2d(p):
    return p.detach().reshape(p.shape[0], -1)


def _splits(n_out, k_out, m_red):
    tiles = ((n_out + 127) // 128) * max(k_out // 64, 1)
    steps = m_red // 32
    s = 1
    while tiles * s < 768 and steps % (2 * s) == 0 and steps // (2 * s) >= 4 and s < 256:
        s *= 2
    return s


def grad_weight(dy_t, x_t, m_red):
    """dy_t [N, M], x_t [K, M] (M contiguous) -> dW [N, K] = sum_m dy[m, n] x[m, k]."""
    n_out, k_out = dy_t.shape[0], x_t.shape[0]
    if m_red % 32:
        raise ValueError("training path needs the number of rows (%d) to be a multiple of 32" % m_red)
    s = _splits(n_out, k_out, m_red)
    ks = m_red // s
    out = torch.empty(n_out, k_out, device=dy_t.device, dtype=torch.float32)
    if s == 1:
        ops.gemm(dy_t, n_out, k_out, ks, [x_t], out, lda=m_red, ldw=m_red)
        return out
    parts = torch.empty(s, n_out, k_out, device=dy_t.device, dtype=torch.float32)
    ops.gemm(dy_t, n_out, k_out, ks, [x_t], parts, lda=m_red, ldw=m_red, groups=s, a_gstride=ks, w_gstride=ks,
             o_gstride=n_out * k_out)
    return ops.reduce_partials(parts, s, n_out * k_out, out)


class _Rows:
    """A row-major [M, N] activation / gradient whose transpose and column sums are produced on demand, once."""

    def __init__(self, x):
        self.x = x
        self._t = None
        self._cs = None

    def t(self):
        if self._t is None:
            if self._cs is None:
                self._t, self._cs = _T_colsum(self.x)
            else:
                self._t = _T(self.x)
        return self._t

    def colsum(self):
        if self._cs is None:
            if self._t is None:
                self._cs = ops.colsum(self.x, self.x.shape[0], self.x.shape[1])
            else:                                   # cannot happen with the call order below; kept for completeness
                self._t, self._cs = _T_colsum(self.x)
        return self._cs


def _tn_splits(n_out, k_out, m_red):
    """Splits of the pixel reduction for ldm_gemm_tn_f32.  (ops.tn_splits_one_round -- one full round of the 512 workgroup slots instead
    of 768 = 1.5 -- was measured on these shapes and is NOT faster: 67.6 vs 65.7 ms of TN time per fp32 step; it pays only for the
    dense 3x3 weight gradients, whose 9 x 2^k tiles land on 576 workgroups.)"""
    tiles = (n_out // 128) * (k_out // 128)
    s = 1
    while tiles * s < 512 and m_red % (2 * s) == 0 and (m_red // (2 * s)) % 32 == 0 and m_red // (2 * s) >= 256 and s < 256:
        s *= 2
    return s


def _gconv_splits(groups, m_red):
    """Pixel splits of the grouped-conv weight-gradient kernel: fill ~512 workgroups, runs of a multiple of 128 pixels."""
    if m_red % 128:
        return 0
    s = 1
    while groups * s < 512 and m_red % (2 * s * 128) == 0 and m_red // (2 * s) >= 512 and s < 128:
        s *= 2
    return s


def grad_weight_rows(dy, x, m_red):
    """dW [N, K] = dy^T x for row-major dy [M, N], x [M, K] (``_Rows``).  Wide layers go through the TN kernel with
    the operands as they lie in memory; narrow ones (tiny test nets) fall back to explicit transposes + the NT GEMM."""
    n_out, k_out = dy.x.shape[1], x.x.shape[1]
    if m_red % 32:
        raise ValueError("training path needs the number of rows (%d) to be a multiple of 32" % m_red)
    if n_out % 128 or k_out % 128:
        return grad_weight(dy.t(), x.t(), m_red)
    s = _tn_splits(n_out, k_out, m_red)
    dev = dy.x.device
    out = torch.empty(n_out, k_out, device=dev, dtype=torch.float32)
    want_cs = dy._cs is None                                    # the bias gradient comes along for free
    cs = torch.empty(s, n_out, device=dev, dtype=torch.float32) if want_cs else None
    if s == 1:
        ops.gemm_tn(dy.x, x.x, out, m_red, n_out, k_out, 1, colsum=cs)
    else:
        parts = torch.empty(s, n_out, k_out, device=dev, dtype=torch.float32)
        ops.gemm_tn(dy.x, x.x, parts, m_red, n_out, k_out, s, colsum=cs)
        if want_cs:                                             # both sums in one launch
            dy._cs = torch.empty(n_out, device=dev, dtype=torch.float32)
            ops.reduce_partials_pair(parts, n_out * k_out, out, cs, n_out, dy._cs, s)
            return out
        ops.reduce_partials(parts, s, n_out * k_out, out)
    if want_cs:
        dy._cs = cs[0]
    return out


class Grads:
    """parameter -> gradient accumulator for one backward."""

    def __init__(self):
        self.g = {}
        self.frozen = set()              # parameters whose gradient has already gone to the bucketed all-reduce

    def add(self, param, grad):
        grad = grad.reshape(param.shape)
        if param in self.frozen:
            raise RuntimeError("a parameter shared between UNet levels received a gradient after its level was handed to the all-reduce")
        if param in self.g:
            ops.add_(self.g[param], grad.contiguous())
        else:
            self.g[param] = grad.contiguous()


# ------------------------------------------------------------------------------------------------------
# SwinBlock
# ------------------------------------------------------------------------------------------------------
def block_forward(blk, rows, shape, ctx, picks, film, codes, enc_hidden):
    """unet.py:42-47 with everything the backward needs kept alive."""
    b, h, w = shape
    m, c = rows.shape
    dev = rows.device
    xf = torch.empty_like(rows)
    ops.channelnorm_film(rows, film, ctx.slot, xf, b, h * w, c, blk.norm.eps)
    y = torch.empty_like(rows)
    ops.gemm(xf, m, 32, 288, [blk._conv_weight()], y, lda=c, ldw=288, biases=[blk.conv.bias.detach()], addend=rows, ldadd=c,
             ldo=c, a_mode=ops.A_CONV3X3, conv_hw=(h, w), cin=32, groups=c // 32, a_gstride=32, w_gstride=32 * 288,
             o_gstride=32, b_gstride=32)
    sv = dict(blk=blk, x=rows, xf=xf, film=film, codes=codes, enc_hidden=enc_hidden, picks=picks, shape=shape)
    if blk.attention_flag:
        att = blk.self_attention.attention
        qkv = torch.empty(m, 3 * c, device=dev, dtype=torch.float32)
        ops.gemm(xf, m, 3 * c, c, [att.in_proj_weight.detach()], qkv, biases=[att.in_proj_bias.detach()])
        actx = torch.empty(m, c, device=dev, dtype=torch.float32)
        ops.window_attention(qkv, att.in_proj_bias.detach(), xf, actx, b, h, w, c, blk.self_attention.window_size,
                             blk.self_attention.shift)
        ops.gemm(actx, m, c, c, [att.out_proj.weight.detach()], y, biases=[att.out_proj.bias.detach()], addend=y)
        sv.update(qkv=qkv, actx=actx)
    regs = [blk.ffn.general] + [blk.ffn.experts[i] for i in picks]
    f = regs[0].a.weight.shape[0]
    a_pre = torch.empty(m, 3 * f, device=dev, dtype=torch.float32)
    b_pre = torch.empty(m, 3 * f, device=dev, dtype=torch.float32)
    hid = torch.empty(m, 3 * f, device=dev, dtype=torch.float32)
    # a(x) * relu(b(x)) of the three ReGLUs with both pre-activations kept for the backward: one launch where the ring kernel takes the shape
    ops.gemm_gate_fwd(xf, m, 3 * f, c, [_w2d(r.a.weight) for r in regs], [_w2d(r.b.weight) for r in regs], hid, a_pre, b_pre,
                      biases_a=[r.a.bias.detach() for r in regs], biases_b=[r.b.bias.detach() for r in regs])
    ops.gemm(hid, m, c, 3 * f, [_w2d(r.c.weight) for r in regs], y, biases=[r.c.bias.detach() for r in regs],
             seg_mode=ops.SEG_K, addend=y)
    sv.update(regs=regs, a_pre=a_pre, b_pre=b_pre, hid=hid)
    return y, sv


def block_backward(sv, dy, ctx, grads):
    blk, x, xf, shape = sv["blk"], sv["x"], sv["xf"], sv["shape"]
    b, h, w = shape
    m, c = x.shape
    dev = x.device
    regs = sv["regs"]
    f = regs[0].a.weight.shape[0]
    dy_r, xf_r = _Rows(dy), _Rows(xf)
    # ---- RandomMoE: y += sum_e c_e(a_e(xf) * relu(b_e(xf))) ----------------------------------------------
    dwc = grad_weight_rows(dy_r, _Rows(sv["hid"]), m)               # [C, 3F]
    bias_dy = dy_r.colsum()
    da = torch.empty(m, 3 * f, device=dev, dtype=torch.float32)
    db = torch.empty_like(da)
    # dhid = dy . Wc with the gate's backward in the epilogue (dhid never reaches HBM) where the ring kernel takes the shape
    ops.gemm_gate_bwd(dy, m, 3 * f, c, [WT.get(_w2d(r.c.weight), r.c.weight) for r in regs], sv["a_pre"], sv["b_pre"], da, db)
    dxf = torch.empty(m, c, device=dev, dtype=torch.float32)
    ops.gemm(da, m, c, 3 * f, [WT.get(_w2d(r.a.weight), r.a.weight) for r in regs], dxf, seg_mode=ops.SEG_K)
    ops.gemm(db, m, c, 3 * f, [WT.get(_w2d(r.b.weight), r.b.weight) for r in regs], dxf, seg_mode=ops.SEG_K, addend=dxf)
    da_r, db_r = _Rows(da), _Rows(db)
    dwa = grad_weight_rows(da_r, xf_r, m)                           # [3F, C]
    dwb = grad_weight_rows(db_r, xf_r, m)
    dba, dbb = da_r.colsum(), db_r.colsum()
    for e, r in enumerate(regs):
        grads.add(r.c.weight, dwc[:, e * f:(e + 1) * f])
        grads.add(r.c.bias, bias_dy.clone())
        grads.add(r.a.weight, dwa[e * f:(e + 1) * f])
        grads.add(r.b.weight, dwb[e * f:(e + 1) * f])
        grads.add(r.a.bias, dba[e * f:(e + 1) * f])
        grads.add(r.b.bias, dbb[e * f:(e + 1) * f])
    # ---- window attention ------------------------------------------------------------------------------
    if blk.attention_flag:
        att = blk.self_attention.attention
        dctx = torch.empty(m, c, device=dev, dtype=torch.float32)
        ops.gemm(dy, m, c, c, [WT.get(att.out_proj.weight.detach(), att.out_proj.weight)], dctx)
        grads.add(att.out_proj.weight, grad_weight_rows(dy_r, _Rows(sv["actx"]), m))
        grads.add(att.out_proj.bias, bias_dy.clone())
        dqkv = torch.empty(m, 3 * c, device=dev, dtype=torch.float32)
        dpad = torch.empty(3 * c, device=dev, dtype=torch.float32)
        ops.window_attention_bwd(sv["qkv"], att.in_proj_bias.detach(), xf, dctx, dqkv, dpad, b, h, w, c,
                                 blk.self_attention.window_size, blk.self_attention.shift)
        ops.gemm(dqkv, m, c, 3 * c, [WT.get(att.in_proj_weight.detach(), att.in_proj_weight)], dxf, addend=dxf)
        dqkv_r = _Rows(dqkv)
        grads.add(att.in_proj_weight, grad_weight_rows(dqkv_r, xf_r, m))
        grads.add(att.in_proj_bias, ops.add_(dqkv_r.colsum().clone(), dpad))
    # ---- grouped 3x3 conv ------------------------------------------------------------------------------
    g = c // 32
    wconv = blk.conv.weight.detach()                               # [C, 32, 3, 3] = [g, co, ci, ky, kx]
    # data gradient = conv of dy with the spatially flipped, in/out-swapped filter: [g, ci, (flipped tap), co]
    wrot = wconv.reshape(g, 32, 32, 3, 3).flip(3, 4).permute(0, 2, 3, 4, 1).reshape(c, 288).contiguous()
    ops.gemm(dy, m, 32, 288, [wrot], dxf, lda=c, ldw=288, addend=dxf, ldadd=c, ldo=c, a_mode=ops.A_CONV3X3, conv_hw=(h, w),
             cin=32, groups=g, a_gstride=32, w_gstride=32 * 288, o_gstride=32, b_gstride=32)
    dwconv = torch.empty(g, 32, 288, device=dev, dtype=torch.float32)
    sp = _gconv_splits(g, m) if 2 <= w <= 96 else 0
    if sp:                                                          # straight from the row-major activations
        planes = torch.empty(4 * sp, c, 288, device=dev, dtype=torch.float32)
        ops.gconv3x3_wgrad(xf, dy, planes, b, h, w, c, sp)
        ops.reduce_partials(planes, 4 * sp, c * 288, dwconv)
    else:                                                           # odd sizes: transposed im2col + one NT GEMM per group
        dy_t = dy_r.t()
        xcol_t = torch.empty(g, 288, m, device=dev, dtype=torch.float32)
        ops.im2col3x3_t(xf, xcol_t, b, h, w, c)
        for gi in range(g):
            dwconv[gi] = grad_weight(dy_t[gi * 32:(gi + 1) * 32], xcol_t[gi], m)
    grads.add(blk.conv.weight, dwconv.reshape(c, 3, 3, 32).permute(0, 3, 1, 2))
    grads.add(blk.conv.bias, bias_dy.clone())
    # ---- ChannelNorm + FiLM, residual -------------------------------------------------------------------
    film = sv["film"]
    dfilm = torch.empty_like(film) if ctx.unique_slots else torch.zeros_like(film)
    dx = torch.empty_like(x)
    ops.channelnorm_film_bwd(x, film, ctx.slot, dxf, dy, dx, dfilm, b, h * w, c, blk.norm.eps, unique_slots=ctx.unique_slots)
    encodings_backward(blk.encodings, sv["codes"], sv["enc_hidden"], dfilm, grads)
    return dx


class LevelCodes:
    """The sin / cos codes of one UNet level for the training step: position part [HW, C] and time part [B, C] (one timestep per
    sample), generated once and shared by the level's SwinBlocks.  Encodings.proj1 is applied to cat[pe(pixel), te(t_b)]
    (unet.py:18-20), so its pre-activation is separable: P[pixel] + T[b]."""

    def __init__(self, ctx, channels, height, width):
        from . import sinusoidal
        t = ctx.t_unique
        self.b, self.hw, self.c = t.numel(), height * width, channels
        self.pe = sinusoidal.embed(t[:1], height, width, channels)[:, :channels].contiguous()          # [HW, C]
        self.te = sinusoidal.embed(t, 1, 1, channels)[:, channels:].contiguous()                       # [B, C]


def _pad_rows(t, mult=32):
    m = t.shape[0]
    mp = (m + mult - 1) // mult * mult
    if mp == m:
        return t, m
    out = torch.zeros(mp, t.shape[1], device=t.device, dtype=t.dtype)
    out[:m] = t
    return out, mp


def _grad_weight_small(dy, x):
    """dW [N, K] = dy^T x for SHORT fp32 operands (HW or B rows): zero-padded to a multiple of 32 rows."""
    dy, mp = _pad_rows(dy)
    x, _ = _pad_rows(x)
    return grad_weight_rows(_Rows(dy), _Rows(x), mp)


def level_pt_rows(encs, lc):
    """P [G, HW, 4C] and T [G, B, 4C] of the Encodings of ALL executed blocks of one level in two grouped launches (pointer-table
    GEMMs: they depend on the codes and the weights only, never on x) instead of two small launches per block."""
    c, n, g = lc.c, 4 * lc.c, len(encs)
    dev = lc.pe.device
    w1s = [_w2d(e.proj1.weight) for e in encs]
    p_all = torch.empty(g, lc.hw, n, device=dev, dtype=torch.float32)
    ops.gemm(lc.pe, lc.hw, n, c, None, p_all, ldw=2 * c, w_table=ops.pointer_table(w1s), groups=g, a_gstride=0, o_gstride=lc.hw * n)
    t_all = torch.empty(g, lc.b, n, device=dev, dtype=torch.float32)
    ops.gemm(lc.te, lc.b, n, c, None, t_all, ldw=2 * c, w_table=ops.pointer_table([w.reshape(-1)[c:] for w in w1s]),
             bias_table=ops.pointer_table([e.proj1.bias.detach() for e in encs]), groups=g, a_gstride=0, o_gstride=lc.b * n)
    return p_all, t_all


def encodings_forward(enc, lc, bf16=False, pt=None):
    """unet.py:20 for one timestep per sample, keeping the hidden activation: hid = relu(P[pixel] + T[b]) with
    P = W1[:, :C] pe, T = W1[:, C:] te + b1 (fp32, tiny; ``pt``: already computed by level_pt_rows), film = proj2(hid).
    -> (hid [B*HW, 4C] fp32 | bf16, film fp32)."""
    c, n = enc.channels, 4 * enc.channels
    dev = lc.pe.device
    if pt is not None:
        p_rows, t_rows = pt
    else:
        w1 = _w2d(enc.proj1.weight)                                                # [4C, 2C]: position columns | time columns
        p_rows = torch.empty(lc.hw, n, device=dev, dtype=torch.float32)
        ops.gemm(lc.pe, lc.hw, n, c, [w1], p_rows, ldw=2 * c)
        t_rows = torch.empty(lc.b, n, device=dev, dtype=torch.float32)
        ops.gemm(lc.te, lc.b, n, c, [w1.reshape(-1)[c:]], t_rows, ldw=2 * c, biases=[enc.proj1.bias.detach()])
    m = lc.b * lc.hw
    hid = torch.empty(m, n, device=dev, dtype=BF16 if bf16 else torch.float32)
    ops.film_hidden(p_rows, t_rows, hid, lc.b, lc.hw, n)
    # bf16 mode: the per-(sample, pixel) FiLM rows are the proj2 GEMM's bf16 output (written once, read by ChannelNorm forward and
    # backward: 2 bytes instead of 4 per value each time); FILM_ROWS_BF16 = False keeps them fp32
    film = torch.empty(m, 2 * c, device=dev, dtype=BF16 if (bf16 and FILM_ROWS_BF16) else torch.float32)
    if bf16:
        ops.gemm_bf16(hid, m, 2 * c, n, [W16.get(enc.proj2.weight)], film, biases=[enc.proj2.bias.detach()])
    else:
        ops.gemm(hid, m, 2 * c, n, [_w2d(enc.proj2.weight)], film, biases=[enc.proj2.bias.detach()])
    return hid, film


def encodings_backward(enc, lc, hid, dfilm, grads):
    """Gradients of proj2 (GEMM-shaped) and of proj1 through its separable form: dP = sum over samples, dT = sum over pixels of
    dh * (hid > 0) (one pass over dh), then dW1 = [dP^T pe | dT^T te], db1 = sum_b dT."""
    c, n = enc.channels, 4 * enc.channels
    m = lc.b * lc.hw
    dev = hid.device
    if hid.dtype == BF16:
        dw2, db2 = grad_weight_rows16(dfilm, hid, m)
        dh = _e16(m, n, dev=dev)
        ops.gemm_bf16(dfilm, m, n, 2 * c, [W16.get(enc.proj2.weight, True)], dh)
    else:
        dfilm_r = _Rows(dfilm)
        dw2 = grad_weight_rows(dfilm_r, _Rows(hid), m)
        db2 = dfilm_r.colsum()
        dh = torch.empty(m, n, device=dev, dtype=torch.float32)
        ops.gemm(dfilm, m, n, 2 * c, [WT.get(_w2d(enc.proj2.weight), enc.proj2.weight)], dh)
    grads.add(enc.proj2.weight, dw2)
    grads.add(enc.proj2.bias, db2)
    dp, dt = ops.film_hidden_bwd(dh, hid, lc.b, lc.hw, n)
    grads.add(enc.proj1.weight, torch.cat([_grad_weight_small(dp, lc.pe), _grad_weight_small(dt, lc.te)], dim=1))
    grads.add(enc.proj1.bias, ops.colsum(dt, lc.b, n))


# ------------------------------------------------------------------------------------------------------
# bf16 operands (BASELINE cfg 5 names bf16; the reference trains under reduced-precision autocast, train_ldm.py:68,80)
#
# Every 1x1-conv / Linear GEMM of the step -- forward, data gradient, weight gradient -- takes bf16 operands that their
# producers rounded ONCE (round-to-nearest-even) and accumulates in fp32 (v_mfma_f32_32x32x16_bf16).  What stays fp32: the
# residual stream and its gradient (each with a bf16 shadow copy for the GEMMs that read it), FiLM rows, the grouped 3x3
# conv (its three kernels), window attention, stem / head / ch_convs, every parameter gradient and AdamW's master weights.
# ------------------------------------------------------------------------------------------------------
PRECISIONS = ("f32", "bf16")
# bf16 training mode: keep the FiLM rows [B*HW, 2C] as bf16 (the proj2 GEMM's bf16 output; ldm_channelnorm_film16_*).  Measured: 0.8 GB less traffic
# per level-0 block and NO change in the step time (68.32 vs 68.34 ms of kernel time: ChannelNorm's backward is bound by its own latency, not
# by the FiLM read) -- so the extra rounding is not taken by default
FILM_ROWS_BF16 = False
BF16 = torch.bfloat16


def set_precision(net, precision):
    """Operand precision of the training step of ``net`` (a UNet): "f32" (exact-fp32 MFMA, default) or "bf16"."""
    if precision not in PRECISIONS:
        raise ValueError("precision must be one of %r" % (PRECISIONS,))
    net.train_precision = precision


class _Weight16:
    """bf16 copies of the GEMM weights viewed as [N, K]: as they lie (forward) and transposed [K, N] (data gradients).
    Master weights are fp32 and move every optimizer step, so the copies are refreshed once per step -- all of them by ONE
    launch (ldm_multi_cast_bf16) over a device-side job table that is rebuilt only when a parameter's storage changed."""

    def __init__(self):
        self.net_id = None
        self.views = {}                    # id(param) -> (source address, plain view, transposed view)
        self.lone = {}
        self.stamp = None

    @staticmethod
    def _gemm_weights(net):
        out = []
        for blk in [b for l in net.encoder_stages for b in l.stage.blocks] + [b for l in net.decoder_stages for b in l.stage.blocks]:
            for r in [blk.ffn.general] + list(blk.ffn.experts):
                out += [r.a.weight, r.b.weight, r.c.weight]
            out.append(blk.encodings.proj2.weight)
            if blk.attention_flag:
                att = blk.self_attention.attention
                out += [att.in_proj_weight, att.out_proj.weight]
        return out

    def refresh(self, net):
        """Call once per training-step forward: re-casts every weight if any of them changed since the last call."""
        import ctypes
        from . import _lib
        plist = self._gemm_weights(net)
        ptrs = tuple(p.data_ptr() for p in plist)
        stamp = (ptrs, sum(p._version for p in plist), weights.GENERATION[0])
        if self.net_id == id(net) and self.stamp == stamp:
            return
        lib = _lib.load()
        rebuild = self.net_id != id(net) or self.stamp is None or self.stamp[0] != ptrs
        if rebuild:
            dev = plist[0].device
            total = sum(p.numel() for p in plist)
            self.buf = torch.empty(2 * total, device=dev, dtype=BF16)
            self.items = (_lib.CastJob * len(plist))()
            self.views = {}
            off = 0
            for it, p in zip(self.items, plist):
                n, k = p.shape[0], p.numel() // p.shape[0]
                plain = self.buf[off:off + n * k].view(n, k)
                trans = self.buf[off + n * k:off + 2 * n * k].view(k, n)
                off += 2 * n * k
                it.src, it.dst, it.dst_t, it.rows, it.cols = p.data_ptr(), plain.data_ptr(), trans.data_ptr(), n, k
                self.views[id(p)] = (p.data_ptr(), plain, trans)
            self.table = torch.empty(lib.ldm_multi_cast_table_bytes(len(plist)), device=dev, dtype=torch.uint8)
            self.tiles = ctypes.c_longlong(0)
            self.keep = plist
        _lib.check(lib.ldm_multi_cast_bf16(self.items, len(plist), self.table.data_ptr(), int(rebuild), ctypes.byref(self.tiles),
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "ldm_multi_cast_bf16")
        # the grouped conv's two bf16 filter tables (forward, data gradient) of every block: one launch per block
        if rebuild:
            self.conv = {}
        for blk in [b for l in net.encoder_stages for b in l.stage.blocks] + [b for l in net.decoder_stages for b in l.stage.blocks]:
            w = blk.conv.weight
            hit = self.conv.get(id(blk))
            if hit is None or hit[0] != w.data_ptr():
                c = w.shape[0]
                hit = self.conv[id(blk)] = (w.data_ptr(), _e16(c, 288, dev=w.device), _e16(c, 288, dev=w.device))
            ops.gconv_pack_bf16(w.detach(), hit[1], hit[2])
        self.net_id, self.stamp = id(net), stamp

    def conv_tables(self, blk):
        """(forward, data-gradient) bf16 filter tables of ``blk``'s grouped conv: from the step's refresh, or built on the spot
        for a block outside the refreshed network."""
        hit = getattr(self, "conv", {}).get(id(blk))
        w = blk.conv.weight
        if hit is not None and hit[0] == w.data_ptr():
            return hit[1], hit[2]
        c = w.shape[0]
        fwd, rot = _e16(c, 288, dev=w.device), _e16(c, 288, dev=w.device)
        ops.gconv_pack_bf16(w.detach().contiguous(), fwd, rot)
        return fwd, rot

    def get(self, p, transposed=False):
        hit = self.views.get(id(p))
        if hit is not None and hit[0] == p.data_ptr():                          # a view into the step's table
            return hit[1 + int(transposed)]
        # a parameter outside the refreshed network (module-level use, tests): one cast per parameter version
        key = weights.key(p)
        lone = self.lone.get(id(p))
        if lone is None or lone[0] != key or lone[3]() is not p:        # see _WeightT.get: entries belong to one live object
            w = _w2d(p).contiguous()
            pid = id(p)
            lone = self.lone[pid] = (key, ops.cast_bf16(w), ops.transpose_cast_bf16(w),
                                     weakref.ref(p, lambda _r, pid=pid, c=self.lone: c.pop(pid, None)))
        return lone[1 + int(transposed)]


W16 = _Weight16()


def _e16(*shape, dev):
    return torch.empty(*shape, device=dev, dtype=BF16)


def _uncast(x16):
    """bf16 -> fp32 copy (narrow test layers only: the fp32 weight-gradient fallback)."""
    out = torch.empty(x16.shape, device=x16.device, dtype=torch.float32)
    ops.uncast_bf16(x16, out)
    return out


import os as _os
# workgroups the split of the pixel reduction aims for: one per CU -- more splits mean more fp32 partial planes to write and sum
# (tiles x splits x tile bytes), which at C >= 512 costs as much as the GEMM itself (measured: 192-384 equal, 512 +2 %, 1024 +10 % step time)
_TN16_TARGET = int(_os.environ.get("LDM_TN16_TARGET", "256"))


def tn16_splits(n_out, k_out, m_red):
    """Splits of the pixel reduction for ldm_gemm_tn_bf16: fill ~512 workgroups of 128 x 256 (K %% 256 == 0) or 128 x 128 tiles."""
    if n_out % 256 == 0 and k_out % 256 == 0:                    # 256 x 256 tiles, one workgroup per CU (gemm_tn_bf16_ring_kernel):
        tiles = (n_out // 256) * (k_out // 256)                    # ONE round of at most 256 workgroups (1.5 rounds ran slower than the 128-row kernel)
        s = 1
        while tiles * s * 2 <= _TN16_TARGET and m_red % (2 * s) == 0 and (m_red // (2 * s)) % 64 == 0 and m_red // (2 * s) >= 256 and s < 256:
            s *= 2
        if tiles * s * 4 >= _TN16_TARGET * 3:                    # >= 3/4 of the CUs busy: the library takes the ring kernel
            return s
    tiles = (n_out // 128) * (k_out // 256 if k_out % 256 == 0 else k_out // 128)
    s = 1
    while tiles * s < _TN16_TARGET and m_red % (2 * s) == 0 and (m_red // (2 * s)) % 64 == 0 and m_red // (2 * s) >= 256 and s < 256:
        s *= 2
    return s


def grad_weight_rows16(dy16, x16, m_red, want_colsum=True, col_blocks=1):
    """(dW [N, K] fp32, column sums of dy [N] fp32 or None) for bf16 row-major dy [M, N], x [M, K]: the TN bf16 kernel with the
    operands as they lie in memory; layers it does not cover (N or K not a multiple of 128, M not a multiple of 64: tiny test
    nets) are up-cast and take the fp32 route.  ``col_blocks`` > 1: dW comes back as that many [N, K / col_blocks] column blocks
    (gradients of separate parameters) -- contiguous tensors straight out of the split sum where there is one, views otherwise."""
    n_out, k_out = dy16.shape[1], x16.shape[1]
    dev = dy16.device

    def blocks(dw):
        if col_blocks == 1:
            return dw
        kb = k_out // col_blocks
        return [dw[:, e * kb:(e + 1) * kb] for e in range(col_blocks)]

    if n_out % 128 or k_out % 128 or m_red % 64:
        dy_r = _Rows(_uncast(dy16))
        dw = grad_weight_rows(dy_r, _Rows(_uncast(x16)), m_red)
        return blocks(dw), (dy_r.colsum() if want_colsum else None)
    s = tn16_splits(n_out, k_out, m_red)
    out = torch.empty(n_out, k_out, device=dev, dtype=torch.float32)
    cs = torch.empty(s, n_out, device=dev, dtype=torch.float32) if want_colsum else None
    if s == 1:
        ops.gemm_tn_bf16(dy16, x16, out, m_red, n_out, k_out, 1, colsum=cs)
        return blocks(out), (cs[0] if want_colsum else None)
    parts = torch.empty(s, n_out, k_out, device=dev, dtype=torch.float32)
    ops.gemm_tn_bf16(dy16, x16, parts, m_red, n_out, k_out, s, colsum=cs)
    if want_colsum:                                             # both sums in one launch
        csum = torch.empty(n_out, device=dev, dtype=torch.float32)
        if col_blocks > 1 and (k_out // col_blocks) % 4 == 0:
            out3 = out.view(col_blocks, n_out, k_out // col_blocks)
            ops.reduce_partials_pair(parts, n_out * k_out, out3, cs, n_out, csum, s, row_len_a=k_out, seg_len_a=k_out // col_blocks)
            return [out3[e] for e in range(col_blocks)], csum
        ops.reduce_partials_pair(parts, n_out * k_out, out, cs, n_out, csum, s)
        return blocks(out), csum
    ops.reduce_partials(parts, s, n_out * k_out, out)
    return blocks(out), None


def block_forward16(blk, rows, shape, ctx, picks, film, codes16, enc_hidden16):
    """block_forward with bf16 GEMM operands; ``rows`` (the residual stream) and the result stay fp32."""
    b, h, w = shape
    m, c = rows.shape
    dev = rows.device
    # window attention reads the bf16 copy too (q, k, v rows and the float "mask" of shifted windows): no fp32 copy is kept
    xf = None
    xf16 = _e16(m, c, dev=dev)
    ops.channelnorm_film_bf16(rows, film, ctx.slot, None, xf16, b, h * w, c, blk.norm.eps)
    y = torch.empty_like(rows)
    ops.gconv3x3_bf16(xf16, W16.conv_tables(blk)[0], blk.conv.bias.detach(), rows, y, b, h, w, c)
    sv = dict(blk=blk, x=rows, xf=xf, xf16=xf16, film=film, codes=codes16, enc_hidden=enc_hidden16, picks=picks, shape=shape)
    if blk.attention_flag:
        att = blk.self_attention.attention
        # q, k, v and the attention context travel as bf16 rows (half the bytes each way); the attention arithmetic itself is fp32
        qkv = _e16(m, 3 * c, dev=dev)
        ops.gemm_bf16(xf16, m, 3 * c, c, [W16.get(att.in_proj_weight)], qkv, biases=[att.in_proj_bias.detach()])
        actx16 = _e16(m, c, dev=dev)
        ops.window_attention_bf16io(qkv, att.in_proj_bias.detach(), xf16, actx16, b, h, w, c, blk.self_attention.window_size,
                                    blk.self_attention.shift)
        ops.gemm_bf16(actx16, m, c, c, [W16.get(att.out_proj.weight)], y, biases=[att.out_proj.bias.detach()], addend=y)
        sv.update(qkv=qkv, actx16=actx16)
    regs = [blk.ffn.general] + [blk.ffn.experts[i] for i in picks]
    f = regs[0].a.weight.shape[0]
    a_pre, b_pre, hid = _e16(m, 3 * f, dev=dev), _e16(m, 3 * f, dev=dev), _e16(m, 3 * f, dev=dev)
    # a(x) * relu(b(x)) of the three ReGLUs in ONE launch that also stores both pre-activations for the backward
    ops.gemm_bf16_gate_fwd(xf16, m, 3 * f, c, [W16.get(r.a.weight) for r in regs], [W16.get(r.b.weight) for r in regs], hid,
                           biases_a=[r.a.bias.detach() for r in regs], biases_b=[r.b.bias.detach() for r in regs], a_pre=a_pre, b_pre=b_pre)
    ops.gemm_bf16(hid, m, c, 3 * f, [W16.get(r.c.weight) for r in regs], y, biases=[r.c.bias.detach() for r in regs],
                  seg_mode=ops.SEG_K, addend=y)
    sv.update(regs=regs, a_pre=a_pre, b_pre=b_pre, hid=hid)
    return y, sv


def block_backward16(sv, dy, dy16, ctx, grads):
    """-> (dx fp32, dx as bf16).  dy16 is the bf16 shadow of dy (the GEMM operand)."""
    blk, x, xf, xf16, shape = sv["blk"], sv["x"], sv["xf"], sv["xf16"], sv["shape"]
    b, h, w = shape
    m, c = x.shape
    dev = x.device
    regs = sv["regs"]
    f = regs[0].a.weight.shape[0]
    # ---- RandomMoE ------------------------------------------------------------------------------------
    dwc, bias_dy = grad_weight_rows16(dy16, sv["hid"], m, col_blocks=3)   # three [C, F] blocks of [C, 3F], [C]
    da, db = _e16(m, 3 * f, dev=dev), _e16(m, 3 * f, dev=dev)
    # dhid = dy . Wc with the gate's backward in the epilogue: dhid never reaches HBM
    ops.gemm_bf16_gate_bwd(dy16, m, 3 * f, c, [W16.get(r.c.weight, True) for r in regs], sv["a_pre"], sv["b_pre"], da, db)
    dxf = torch.empty(m, c, device=dev, dtype=torch.float32)
    ops.gemm_bf16(da, m, c, 3 * f, [W16.get(r.a.weight, True) for r in regs], dxf, seg_mode=ops.SEG_K)
    ops.gemm_bf16(db, m, c, 3 * f, [W16.get(r.b.weight, True) for r in regs], dxf, seg_mode=ops.SEG_K, addend=dxf)
    dwa, dba = grad_weight_rows16(da, xf16, m)                        # [3F, C]
    dwb, dbb = grad_weight_rows16(db, xf16, m)
    # the column sums of dy are the gradient of every bias added to y: three c-biases, the conv bias, the out-projection bias --
    # separate rows (one launch) so that no two parameters share gradient storage
    brows = ops.replicate(bias_dy, 4 + int(blk.attention_flag))
    for e, r in enumerate(regs):
        grads.add(r.c.weight, dwc[e])
        grads.add(r.c.bias, brows[e])
        grads.add(r.a.weight, dwa[e * f:(e + 1) * f])
        grads.add(r.b.weight, dwb[e * f:(e + 1) * f])
        grads.add(r.a.bias, dba[e * f:(e + 1) * f])
        grads.add(r.b.bias, dbb[e * f:(e + 1) * f])
    # ---- window attention ---------------------------------------------------------------------------
    if blk.attention_flag:
        att = blk.self_attention.attention
        dctx = _e16(m, c, dev=dev)
        ops.gemm_bf16(dy16, m, c, c, [W16.get(att.out_proj.weight, True)], dctx)
        dwo, _ = grad_weight_rows16(dy16, sv["actx16"], m, want_colsum=False)
        grads.add(att.out_proj.weight, dwo)
        grads.add(att.out_proj.bias, brows[4])
        dqkv16 = _e16(m, 3 * c, dev=dev)
        dpad = torch.empty(3 * c, device=dev, dtype=torch.float32)
        ops.window_attention_bwd_bf16(sv["qkv"], att.in_proj_bias.detach(), xf16, dctx, dqkv16, dpad, b, h, w, c,
                                      blk.self_attention.window_size, blk.self_attention.shift)
        ops.gemm_bf16(dqkv16, m, c, 3 * c, [W16.get(att.in_proj_weight, True)], dxf, addend=dxf)
        dwi, dbi = grad_weight_rows16(dqkv16, xf16, m)
        grads.add(att.in_proj_weight, dwi)
        grads.add(att.in_proj_bias, ops.add_(dbi.clone(), dpad))
    # ---- grouped 3x3 conv: data gradient on the bf16 matrix cores, weight gradient by the fp32 kernel ----------
    _gconv_backward(blk, xf, dy, dxf, brows[3], shape, grads, dy16=dy16, xf16=xf16)
    # ---- ChannelNorm + FiLM, residual ---------------------------------------------------------------------
    film = sv["film"]
    dfilm16 = _e16(film.shape[0], film.shape[1], dev=dev)
    dx = torch.empty_like(x)
    dx16 = _e16(m, c, dev=dev)
    ops.channelnorm_film_bwd_bf16(x, film, ctx.slot, dxf, dy, dx, dx16, dfilm16, b, h * w, c, blk.norm.eps)
    encodings_backward(blk.encodings, sv["codes"], sv["enc_hidden"], dfilm16, grads)
    return dx, dx16


def _gconv_backward(blk, xf, dy, dxf, bias_dy, shape, grads, dy16=None, xf16=None):
    """data gradient (accumulated into dxf) and weight / bias gradient of the grouped 3x3 conv (unet.py:30,44); both take bf16
    operands when ``dy16`` / ``xf16`` (the bf16 shadows of dy and of the block's normalised input) are given."""
    b, h, w = shape
    m, c = dy.shape
    dev = dy.device
    g = c // 32
    wconv = blk.conv.weight.detach()                               # [C, 32, 3, 3] = [g, co, ci, ky, kx]
    if dy16 is not None:
        ops.gconv3x3_bf16(dy16, W16.conv_tables(blk)[1], None, dxf, dxf, b, h, w, c)
        dwconv = ops.gconv3x3_wgrad_bf16(xf16, dy16, b, h, w, c)
        grads.add(blk.conv.weight, dwconv.reshape(c, 3, 3, 32).permute(0, 3, 1, 2))
        grads.add(blk.conv.bias, bias_dy)                          # the caller's own row of the replicated column sums
        return
    else:
        wrot = wconv.reshape(g, 32, 32, 3, 3).flip(3, 4).permute(0, 2, 3, 4, 1).reshape(c, 288).contiguous()
        ops.gemm(dy, m, 32, 288, [wrot], dxf, lda=c, ldw=288, addend=dxf, ldadd=c, ldo=c, a_mode=ops.A_CONV3X3, conv_hw=(h, w),
                 cin=32, groups=g, a_gstride=32, w_gstride=32 * 288, o_gstride=32, b_gstride=32)
    dwconv = torch.empty(g, 32, 288, device=dev, dtype=torch.float32)
    sp = _gconv_splits(g, m) if 2 <= w <= 96 else 0
    if sp:
        planes = torch.empty(4 * sp, c, 288, device=dev, dtype=torch.float32)
        ops.gconv3x3_wgrad(xf, dy, planes, b, h, w, c, sp)
        ops.reduce_partials(planes, 4 * sp, c * 288, dwconv)
    else:
        dy_t = _T(dy)
        xcol_t = torch.empty(g, 288, m, device=dev, dtype=torch.float32)
        ops.im2col3x3_t(xf, xcol_t, b, h, w, c)
        for gi in range(g):
            dwconv[gi] = grad_weight(dy_t[gi * 32:(gi + 1) * 32], xcol_t[gi], m)
    grads.add(blk.conv.weight, dwconv.reshape(c, 3, 3, 32).permute(0, 3, 1, 2))
    grads.add(blk.conv.bias, bias_dy.clone())


# ------------------------------------------------------------------------------------------------------
# whole UNet
# ------------------------------------------------------------------------------------------------------
class UNetFunction(torch.autograd.Function):
    """unet.py:89-103 forward + backward on the HIP path.  ``params`` only anchors the graph."""

    @staticmethod
    def forward(fctx, net, x, time, *params):
        from .unet import TimeContext
        b, cin, h, w = x.shape
        dev = x.device
        ctx = TimeContext(time, b, dev, dedupe=False)
        order = [blk for l in net.encoder_stages for blk in l.stage.blocks] + [blk for l in net.decoder_stages for blk in l.stage.blocks]
        decisions = {blk: blk.draw() for blk in order}
        tape = []
        c0 = net.channels[0]
        x = x.contiguous().float()
        rows = torch.empty(b * h * w, c0, device=dev, dtype=torch.float32)
        ops.stem_nchw(x, _w2d(net.encoder_first.weight), net.encoder_first.bias.detach(), rows, b, cin, h * w, c0)

        bf16 = getattr(net, "train_precision", "f32") == "bf16"
        if bf16 and any(ch % 64 for ch in net.channels):
            raise ValueError("bf16 training needs every stage width to be a multiple of 64 (got %r)" % (net.channels,))
        if bf16:
            W16.refresh(net)
        level_codes = {}
        # the Encodings' P / T rows of every executed block, level by level, up front (two grouped launches per level)
        pt_rows = {}
        for i, ch in enumerate(net.channels):
            live = [blk for blk in net._level_blocks(i) if decisions[blk] is not None]
            # bf16 mode only: the grouped launches do not take the split-K route the per-block fp32 launches take at M <= 128, i.e. they
            # re-associate the fp32 sums -- harmless (5e-7), but in exact-fp32 mode the full-width goldens are held to 2e-4 per gradient
            # norm, and one ReLU gate of a 128-row level flipping against the reference moves a norm by more than that
            if bf16 and live and (h >> i) >= 1 and (w >> i) >= 1 and len(live) <= 32:
                key = (ch, h >> i, w >> i)
                level_codes[key] = LevelCodes(ctx, *key)
                p_all, t_all = level_pt_rows([blk.encodings for blk in live], level_codes[key])
                for k, blk in enumerate(live):
                    pt_rows[blk] = (p_all[k], t_all[k])

        def run_stage(stage, rows, shape):
            for blk in stage.blocks:
                picks = decisions[blk]
                if picks is None:
                    continue
                key = (rows.shape[1], shape[1], shape[2])
                if key not in level_codes:                      # one code table per level, shared by its blocks
                    level_codes[key] = LevelCodes(ctx, *key)
                lc = level_codes[key]
                enc_hidden, film = encodings_forward(blk.encodings, lc, bf16, pt=pt_rows.get(blk))
                fwd = block_forward16 if bf16 else block_forward
                rows, sv = fwd(blk, rows, shape, ctx, picks, film, lc, enc_hidden)
                tape.append(("block", sv))
            return rows

        skips = []
        n = len(net.encoder_stages)
        for i, l in enumerate(net.encoder_stages):
            rows = run_stage(l.stage, rows, (b, h, w))
            if i == n - 1:
                skips.insert(0, None)
            else:
                skips.insert(0, rows)
                conv = l.ch_conv[0]
                pooled = torch.empty(b * (h // 2) * (w // 2), rows.shape[1], device=dev, dtype=torch.float32)
                ops.avgpool2(rows, pooled, b, h, w, rows.shape[1])
                h, w = h // 2, w // 2
                rows = torch.empty(b * h * w, conv.weight.shape[0], device=dev, dtype=torch.float32)
                ops.gemm(pooled, b * h * w, conv.weight.shape[0], pooled.shape[1], [_w2d(conv.weight)], rows, biases=[conv.bias.detach()])
                tape.append(("down", dict(conv=conv, pooled=pooled, shape=(b, h, w), level=i)))
        for j, (l, s) in enumerate(zip(net.decoder_stages, skips)):
            if not isinstance(l.ch_conv, torch.nn.Identity):
                conv = l.ch_conv[1]
                up = torch.empty(b * 4 * h * w, conv.weight.shape[0], device=dev, dtype=torch.float32)
                ops.gemm(rows, b * h * w, conv.weight.shape[0], rows.shape[1], [_w2d(conv.weight)], up, biases=[conv.bias.detach()],
                         addend=s, o_mode=ops.O_UP2, out_hw=(h, w))
                tape.append(("up", dict(conv=conv, lo=rows, shape=(b, h, w), level=n - 1 - j)))
                h, w = 2 * h, 2 * w
                rows = up
            rows = run_stage(l.stage, rows, (b, h, w))
        out = torch.empty(b, cin, h, w, device=dev, dtype=torch.float32)
        wl = net.decoder_last.weight.detach().reshape(c0, cin)
        ops.head_nchw(rows, wl, net._head_bias(), out, b, c0, h * w, cin)
        fctx.net, fctx.tape, fctx.tctx, fctx.x, fctx.last_rows, fctx.params, fctx.bf16 = net, tape, ctx, x, rows, params, bf16
        return out

    @staticmethod
    def backward(fctx, dout):
        net, tape, ctx, x, params = fctx.net, fctx.tape, fctx.tctx, fctx.x, fctx.params
        b, cin, h, w = x.shape
        dev = x.device
        grads = Grads()
        c0 = net.channels[0]
        rows = fctx.last_rows
        drows = torch.empty_like(rows)
        dwl = torch.empty(c0, cin, device=dev, dtype=torch.float32)
        dbl = torch.empty(cin, device=dev, dtype=torch.float32)
        ops.head_bwd(rows, net.decoder_last.weight.detach().reshape(c0, cin), dout.contiguous().float(), drows, dwl, dbl, b, c0, h * w, cin)
        grads.add(net.decoder_last.weight, dwl)
        ss = net.stem_size * net.stem_size                 # stem_size > 1: the head bias was replicated over the s x s patch positions
        grads.add(net.decoder_last.bias, dbl if ss == 1 else dbl.reshape(cin // ss, ss).sum(1))
        dskip = {}
        drows16 = None                                       # bf16 shadow of drows (bf16 mode): produced by the block that wrote drows
        # data-parallel step: hand each finished level's gradients to the bucketed all-reduce while the next level computes
        sync = getattr(net, "_grad_sync", None)
        flushed = set()

        def flush():
            if sync is None:
                return
            items = [(p, g) for p, g in grads.g.items() if p not in flushed]
            flushed.update(p for p, _ in items)
            grads.frozen = flushed
            sync.push(items, grads.g)

        for kind, sv in reversed(tape):
            if kind != "block":
                flush()                                      # a level boundary (ch_conv): everything above it is final
            if kind == "block":
                if fctx.bf16:
                    if drows16 is None:
                        drows16 = ops.cast_bf16(drows)
                    drows, drows16 = block_backward16(sv, drows, drows16, ctx, grads)
                else:
                    drows = block_backward(sv, drows, ctx, grads)
                continue
            drows16 = None
            if kind == "block":
                pass
            elif kind == "up":
                conv, lo, (bb, lh, lw) = sv["conv"], sv["lo"], sv["shape"]
                cn = conv.weight.shape[0]
                dskip[sv["level"]] = drows                                  # gradient of the additive skip (unet.py:101)
                dlo = torch.empty(bb * lh * lw, cn, device=dev, dtype=torch.float32)
                ops.sumpool2(drows, dlo, bb, 2 * lh, 2 * lw, cn)
                mlo = bb * lh * lw
                dlo_r = _Rows(dlo)
                grads.add(conv.weight, grad_weight_rows(dlo_r, _Rows(lo), mlo))
                grads.add(conv.bias, dlo_r.colsum())
                drows = torch.empty(mlo, lo.shape[1], device=dev, dtype=torch.float32)
                ops.gemm(dlo, mlo, lo.shape[1], cn, [WT.get(_w2d(conv.weight), conv.weight)], drows)
            elif kind == "down":
                conv, pooled, (bb, lh, lw) = sv["conv"], sv["pooled"], sv["shape"]
                mlo = bb * lh * lw
                dr_r = _Rows(drows)
                grads.add(conv.weight, grad_weight_rows(dr_r, _Rows(pooled), mlo))
                grads.add(conv.bias, dr_r.colsum())
                dpool = torch.empty(mlo, pooled.shape[1], device=dev, dtype=torch.float32)
                ops.gemm(drows, mlo, pooled.shape[1], conv.weight.shape[0], [WT.get(_w2d(conv.weight), conv.weight)], dpool)
                hi = dskip.pop(sv["level"]).clone()                         # start from the skip gradient, add the pooled path
                ops.avgpool2_bwd(dpool, hi, bb, 2 * lh, 2 * lw, pooled.shape[1], True)
                drows = hi
        dw0 = torch.empty(c0, cin, device=dev, dtype=torch.float32)
        ops.stem_bwd(x, drows, dw0, b, cin, h * w, c0)
        grads.add(net.encoder_first.weight, dw0)
        grads.add(net.encoder_first.bias, ops.colsum(drows, b * h * w, c0))
        dx = None
        if fctx.needs_input_grad[1]:
            # dL/dx of the 1x1 stem (a trainable encoder upstream, gradient-based guidance): drows . W, rows -> NCHW -- the
            # head kernel's arithmetic with the stem weight [C0, Cin]
            dx = torch.empty(b, cin, h, w, device=dev, dtype=torch.float32)
            ops.head_nchw(drows, _w2d(net.encoder_first.weight).contiguous(), None, dx, b, c0, h * w, cin)
        fctx.tape = None
        if sync is not None:
            flush()
            sync.finish()                                    # averaged gradients are written back into grads.g
        # the returned tuple holds the ONLY references to the gradient tensors: autograd's AccumulateGrad then adopts them as .grad
        # instead of cloning each one (190 copies per step at the default widths)
        out = tuple(grads.g.get(p) for p in params)
        grads.g.clear()
        return (None, dx, None) + out


class L1LossFunction(torch.autograd.Function):
    """nn.L1Loss() (mean reduction) of ddpm.py:16,47 as two kernels."""

    @staticmethod
    def forward(fctx, pred, target):
        pred, target = pred.contiguous().float(), target.contiguous().float()
        loss = torch.empty(1, device=pred.device, dtype=torch.float32)
        ops.l1_loss(pred, target, loss)
        fctx.save_for_backward(pred, target)
        return loss.reshape(())

    @staticmethod
    def backward(fctx, gout):
        pred, target = fctx.saved_tensors
        grad = torch.empty_like(pred)
        ops.l1_loss_bwd(pred, target, gout.reshape(1).contiguous().float(), grad)
        return grad, None
