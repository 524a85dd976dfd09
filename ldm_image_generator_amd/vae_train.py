"""Backward of the VAE ``Decoder`` (vae.py:99-132: stem 1x1, ConvTranspose 2x2 up-samplings, ResBlocks of dense 3x3 convs with
leaky ReLU, to_rgb heads accumulated through bilinear x2) on the HIP path -- SURVEY f4 / the VAE training objective's decoder side.

ONE ``torch.autograd.Function`` for the whole decoder, like ``train.UNetFunction``: the forward runs the inference kernels and keeps
the tape (per ResBlock: its input, the activated hidden ``y1`` and the activated branch ``t2``), the backward walks it in reverse:

* dense 3x3 data gradient  = the same implicit-GEMM conv on the flipped, in/out-swapped filter (``ldm_gemm_f32``, ``A_CONV3X3``);
* dense 3x3 weight gradient = ``dy^T . im2col(x)`` (``ldm_im2col3x3_f32`` + the TN / NT weight-gradient GEMMs of ``train.py``);
* ConvTranspose 2x2 (stride 2) = a plain GEMM per fine-pixel parity: gradients through ``ldm_space_to_depth2_f32``;
* leaky ReLU: ``ldm_lrelu_bwd_f32`` on the ACTIVATED outputs (slope > 0 keeps the sign);
* to_rgb + bilinear x2 accumulation: ``ldm_rgb_head_bwd_f32``.

fp32 throughout (the reference trains the VAE under autocast; this is the exact-fp32 statement of the same graph)."""
import torch

from . import ops
from .modules import w2d
from .train import _Rows, grad_weight_rows

SLOPE = 0.01            # F.leaky_relu's default negative_slope (vae.py:62,64)
IMPLICIT_WGRAD = True   # dense 3x3 weight gradients without the im2col matrix (False: explicit im2col + TN / NT GEMM, the round-2 form)


def _conv(rows, shape, packed, bias, cin, cout, act, addend=None):
    b, h, w = shape
    out = torch.empty(rows.shape[0], cout, device=rows.device, dtype=torch.float32)
    ops.gemm(rows, rows.shape[0], cout, 9 * cin, [packed], out, lda=cin, ldw=9 * cin, biases=None if bias is None else [bias],
             act=act, slope=SLOPE, addend=addend, a_mode=ops.A_CONV3X3, conv_hw=(h, w), cin=cin)
    return out


def _conv_grads(dy, x, shape, conv, grads):
    """weight / bias gradient of one dense 3x3 conv from the gradient at its pre-activation and its input rows."""
    b, h, w = shape
    cout, cin = conv.weight.shape[0], conv.weight.shape[1]
    if IMPLICIT_WGRAD and cin % 4 == 0 and cout % 4 == 0 and dy.shape[0] % 32 == 0:
        # implicit im2col inside the weight-gradient GEMM (ldm_conv3x3_wgrad_f32): the explicit matrix is 9x the activation
        # (1.2 GB per conv at 256 x 256, batch 8, C = 64) written once and read once
        dw, db = ops.conv3x3_wgrad(dy, x, b, h, w, cin, cout)
        grads[conv.weight] = dw.reshape(cout, 3, 3, cin).permute(0, 3, 1, 2).contiguous()
        grads[conv.bias] = db.contiguous()
        return
    dyr = _Rows(dy)
    dw = grad_weight_rows(dyr, _Rows(ops.im2col3x3(x, b, h, w, cin)), dy.shape[0])          # [Cout, 9 Cin]
    grads[conv.weight] = dw.reshape(cout, 3, 3, cin).permute(0, 3, 1, 2).contiguous()
    grads[conv.bias] = dyr.colsum().clone()


def _resblock_forward(blk, rows, shape, tape):
    """x + lrelu(c2(lrelu(c1(x))))  (vae.py:60-66) keeping x, the activated hidden and the activated branch."""
    c = rows.shape[1]
    p1, q1 = ops.pack3x3(blk.c1.weight)                  # forward and data-gradient filter matrices: one launch per conv
    p2, q2 = ops.pack3x3(blk.c2.weight)
    y1 = _conv(rows, shape, p1, blk.c1.bias.detach(), c, c, ops.ACT_LRELU)
    t2 = _conv(y1, shape, p2, blk.c2.bias.detach(), c, c, ops.ACT_LRELU)
    tape.append(dict(blk=blk, x=rows, y1=y1, t2=t2, q1=q1, q2=q2))
    return ops.add_(t2.clone(), rows)


def _resblock_backward(rec, drows, shape, grads):
    blk, x, y1, t2 = rec["blk"], rec["x"], rec["y1"], rec["t2"]
    c = x.shape[1]
    d2 = ops.lrelu_bwd(drows, t2, SLOPE)                                  # at c2's pre-activation
    _conv_grads(d2, y1, shape, blk.c2, grads)
    dy1 = _conv(d2, shape, rec["q2"], None, c, c, ops.ACT_NONE)
    d1 = ops.lrelu_bwd(dy1, y1, SLOPE)
    _conv_grads(d1, x, shape, blk.c1, grads)
    return _conv(d1, shape, rec["q1"], None, c, c, ops.ACT_NONE, addend=drows)          # + the skip path


class DecoderFunction(torch.autograd.Function):
    """vae.py:122-132 forward + backward.  ``params`` only anchors the graph."""

    @staticmethod
    def forward(fctx, dec, z, *params):
        from torch import nn
        b, cz, h, w = z.shape
        dev = z.device
        z = z.contiguous().float()
        c0 = dec.input_layer.weight.shape[0]
        rows = torch.empty(b * h * w, c0, device=dev, dtype=torch.float32)
        ops.stem_nchw(z, w2d(dec.input_layer), dec.input_layer.bias.detach(), rows, b, cz, h * w, c0)
        tape, rgb = [], None
        for i, (up, stage) in enumerate(zip(dec.upsamples, dec.stages)):
            rec = dict(up=None, blocks=[])
            if not isinstance(up, nn.Identity):
                cin, cout = up.weight.shape[0], up.weight.shape[1]
                packed = up.weight.detach().permute(2, 3, 1, 0).reshape(4 * cout, cin).contiguous()      # [(dy, dx, co)][ci]
                fine = torch.empty(b * 4 * h * w, cout, device=dev, dtype=torch.float32)
                ops.gemm(rows, b * h * w, 4 * cout, cin, [packed], fine, biases=[up.bias.detach()], ldo=cout, o_mode=ops.O_CONVT2X2,
                         out_hw=(h, w), cout=cout)
                rec["up"] = dict(conv=up, x=rows, packed=packed, hw=(h, w))
                rows, h, w = fine, 2 * h, 2 * w
            shape = (b, h, w)
            for blk in stage.layers:
                rows = _resblock_forward(blk, rows, shape, rec["blocks"])
            new_rgb = torch.empty(b, stage.to_rgb.weight.shape[0], h, w, device=dev, dtype=torch.float32)
            ops.rgb_head(rows, w2d(stage.to_rgb), stage.to_rgb.bias.detach(), rgb, new_rgb, b, h, w, rows.shape[1])
            rec.update(stage=stage, rows=rows, shape=shape, has_prev=rgb is not None)
            tape.append(rec)
            rgb = new_rgb
        fctx.dec, fctx.tape, fctx.z, fctx.params = dec, tape, z, params
        return rgb

    @staticmethod
    def backward(fctx, dout):
        dec, tape, z = fctx.dec, fctx.tape, fctx.z
        dev = dout.device
        grads = {}
        drgb = dout.contiguous().float()
        drows = None                                                  # gradient w.r.t. the stage's output rows coming from the finer stage
        for rec in reversed(tape):
            stage, rows, (b, h, w) = rec["stage"], rec["rows"], rec["shape"]
            c = rows.shape[1]
            oc = stage.to_rgb.weight.shape[0]
            dw_rgb = torch.zeros(oc, c, device=dev, dtype=torch.float32)
            db_rgb = torch.zeros(oc, device=dev, dtype=torch.float32)
            dprev = torch.zeros(b, oc, h // 2, w // 2, device=dev, dtype=torch.float32) if rec["has_prev"] else None
            accumulate = drows is not None
            if drows is None:
                drows = torch.empty_like(rows)
            ops.rgb_head_bwd(drgb, w2d(stage.to_rgb).contiguous(), rows, drows, accumulate, dprev, dw_rgb, db_rgb, b, h, w, c)
            grads[stage.to_rgb.weight] = dw_rgb.reshape(stage.to_rgb.weight.shape)
            grads[stage.to_rgb.bias] = db_rgb
            drgb = dprev
            for blk_rec in reversed(rec["blocks"]):
                drows = _resblock_backward(blk_rec, drows, (b, h, w), grads)
            if rec["up"] is not None:
                up, x_in, packed, (hc, wc) = rec["up"]["conv"], rec["up"]["x"], rec["up"]["packed"], rec["up"]["hw"]
                cin, cout = up.weight.shape[0], up.weight.shape[1]
                d4 = ops.space_to_depth2(drows, b, hc, wc, cout)                       # [M_coarse, 4 Cout]
                d4r = _Rows(d4)
                dwp = grad_weight_rows(d4r, _Rows(x_in), d4.shape[0])                  # [(dy, dx, co)][ci]
                grads[up.weight] = dwp.reshape(2, 2, cout, cin).permute(3, 2, 0, 1).contiguous()
                grads[up.bias] = d4r.colsum().reshape(4, cout).sum(0)
                drows = torch.empty(d4.shape[0], cin, device=dev, dtype=torch.float32)
                ops.gemm(d4, d4.shape[0], cin, 4 * cout, [packed.t().contiguous()], drows)
        # stem (vae.py:124): 1x1 conv from NCHW
        b, cz, h0, w0 = z.shape
        c0 = dec.input_layer.weight.shape[0]
        dw0 = torch.empty(c0, cz, device=dev, dtype=torch.float32)
        ops.stem_bwd(z, drows, dw0, b, cz, h0 * w0, c0)
        grads[dec.input_layer.weight] = dw0.reshape(dec.input_layer.weight.shape)
        grads[dec.input_layer.bias] = ops.colsum(drows, drows.shape[0], c0)
        dz = None
        if fctx.needs_input_grad[1]:
            dz = torch.empty_like(z)
            ops.head_nchw(drows, w2d(dec.input_layer).contiguous(), None, dz, b, c0, h0 * w0, cz)
        return (None, dz) + tuple(grads.get(p) for p in fctx.params)


class EncoderFunction(torch.autograd.Function):
    """vae.py:76-96 forward + backward: stem 1x1 from NCHW, ResBlocks, (avg-pool 2 -> 1x1 conv) between stages, 1x1 head to NCHW."""

    @staticmethod
    def forward(fctx, enc, x, *params):
        from torch import nn
        b, cin, h, w = x.shape
        dev = x.device
        x = x.contiguous().float()
        c0 = enc.input_layer.weight.shape[0]
        rows = torch.empty(b * h * w, c0, device=dev, dtype=torch.float32)
        ops.stem_nchw(x, w2d(enc.input_layer), enc.input_layer.bias.detach(), rows, b, cin, h * w, c0)
        tape = []
        for stage, down in zip(enc.stages, enc.downsamples):
            rec = dict(blocks=[], shape=(b, h, w), down=None)
            for blk in stage.seq:
                rows = _resblock_forward(blk, rows, (b, h, w), rec["blocks"])
            if not isinstance(down, nn.Identity):
                conv = down[1]
                c = rows.shape[1]
                pooled = torch.empty(b * (h // 2) * (w // 2), c, device=dev, dtype=torch.float32)
                ops.avgpool2(rows, pooled, b, h, w, c)
                h, w = h // 2, w // 2
                rows = torch.empty(b * h * w, conv.weight.shape[0], device=dev, dtype=torch.float32)
                ops.gemm(pooled, b * h * w, conv.weight.shape[0], c, [w2d(conv)], rows, biases=[conv.bias.detach()])
                rec["down"] = dict(conv=conv, pooled=pooled)
            tape.append(rec)
        cz = enc.output_layer.weight.shape[0]
        z = torch.empty(b, cz, h, w, device=dev, dtype=torch.float32)
        w_t = enc.output_layer.weight.detach().reshape(cz, -1).t().contiguous()                 # [C, latent]
        ops.head_nchw(rows, w_t, enc.output_layer.bias.detach(), z, b, rows.shape[1], h * w, cz)
        fctx.enc, fctx.tape, fctx.x, fctx.params, fctx.last, fctx.w_t, fctx.hw = enc, tape, x, params, rows, w_t, (h, w)
        return z

    @staticmethod
    def backward(fctx, dz):
        enc, tape, x, rows, w_t, (h, w) = fctx.enc, fctx.tape, fctx.x, fctx.last, fctx.w_t, fctx.hw
        dev = dz.device
        b, cin = x.shape[0], x.shape[1]
        grads = {}
        c, cz = rows.shape[1], enc.output_layer.weight.shape[0]
        drows = torch.empty_like(rows)
        dwl = torch.empty(c, cz, device=dev, dtype=torch.float32)
        dbl = torch.empty(cz, device=dev, dtype=torch.float32)
        ops.head_bwd(rows, w_t, dz.contiguous().float(), drows, dwl, dbl, b, c, h * w, cz)
        grads[enc.output_layer.weight] = dwl.t().reshape(enc.output_layer.weight.shape).contiguous()
        grads[enc.output_layer.bias] = dbl
        for rec in reversed(tape):
            bb, hh, ww = rec["shape"]
            if rec["down"] is not None:                                   # rows_coarse = conv1x1(avgpool2(rows_fine))
                conv, pooled = rec["down"]["conv"], rec["down"]["pooled"]
                cout, cfine = conv.weight.shape[0], conv.weight.shape[1]
                dr = _Rows(drows)
                grads[conv.weight] = grad_weight_rows(dr, _Rows(pooled), drows.shape[0]).reshape(conv.weight.shape)
                grads[conv.bias] = dr.colsum().clone()
                dpooled = torch.empty_like(pooled)
                ops.gemm(drows, drows.shape[0], cfine, cout, [w2d(conv).t().contiguous()], dpooled)
                drows = torch.empty(bb * hh * ww, cfine, device=dev, dtype=torch.float32)
                ops.avgpool2_bwd(dpooled, drows, bb, hh, ww, cfine, False)
            for blk_rec in reversed(rec["blocks"]):
                drows = _resblock_backward(blk_rec, drows, (bb, hh, ww), grads)
        c0 = enc.input_layer.weight.shape[0]
        h0, w0 = x.shape[2], x.shape[3]
        dw0 = torch.empty(c0, cin, device=dev, dtype=torch.float32)
        ops.stem_bwd(x, drows, dw0, b, cin, h0 * w0, c0)
        grads[enc.input_layer.weight] = dw0.reshape(enc.input_layer.weight.shape)
        grads[enc.input_layer.bias] = ops.colsum(drows, drows.shape[0], c0)
        dx = None
        if fctx.needs_input_grad[1]:
            dx = torch.empty_like(x)
            ops.head_nchw(drows, w2d(enc.input_layer).contiguous(), None, dx, b, c0, h0 * w0, cin)
        return (None, dx) + tuple(grads.get(p) for p in fctx.params)


class AddNoiseFunction(torch.autograd.Function):
    """z + noise * gain (vae.py:38) with torch's two roundings; dL/dz passes through."""

    @staticmethod
    def forward(fctx, z, noise, gain):
        b = z.shape[0]
        out = torch.empty_like(z)
        ops.qsample(z.contiguous(), noise, torch.ones(b, device=z.device), torch.full((b,), float(gain), device=z.device), out)
        return out

    @staticmethod
    def backward(fctx, g):
        return g, None, None


class ToRowsFunction(torch.autograd.Function):
    """z [B, C, h, w] -> [B*h*w, C] (== z.reshape(B, C, -1).transpose(1, 2), vae.py:40) and back for the gradient."""

    @staticmethod
    def forward(fctx, z):
        b, c, h, w = z.shape
        fctx.shape = (b, c, h, w)
        rows = torch.empty(b * h * w, c, device=z.device, dtype=torch.float32)
        ops.nchw_to_nhwc(z.contiguous().float(), rows, b, c, h * w)
        return rows

    @staticmethod
    def backward(fctx, g):
        b, c, h, w = fctx.shape
        out = torch.empty(b, c, h, w, device=g.device, dtype=torch.float32)
        ops.nhwc_to_nchw(g.contiguous().float(), out, b, c, h * w)
        return out


# ------------------------------------------------------------------------------------------------------
# Discriminator (vae.py:134-171): stem 1x1, ResStacks at 32 / 48 / 48 / 96 channels, Conv 2x2 stride 2 between stages, and
# one 1-channel "early exit" per stage whose spatial mean is summed into the logit.
# ------------------------------------------------------------------------------------------------------
def _pad32(c):
    """row width a c-channel layer is carried at: the GEMM family works on 32-column multiples, and the dense 3x3 conv on
    Cin % 32 == 0, so the 48-channel stages run at 64 with zero weights / biases in the 16 extra channels.  Those channels stay
    exactly zero through conv + leaky ReLU + skip, and their (meaningless) weight gradients are sliced away."""
    return c if c % 32 == 0 else (c + 63) // 64 * 64


class _PaddedConv:
    """zero-padded stand-in for an nn.Conv2d(c, c', k): ``weight`` [P(c'), P(c), k, k], ``bias`` [P(c')]."""

    def __init__(self, conv):
        w, b = conv.weight.detach(), conv.bias.detach()
        co, ci = w.shape[0], w.shape[1]
        pco, pci = _pad32(co), _pad32(ci)
        self.real = conv
        if (pco, pci) == (co, ci):
            self.weight, self.bias = w, b
        else:
            self.weight = torch.zeros(pco, pci, w.shape[2], w.shape[3], device=w.device, dtype=torch.float32)
            self.weight[:co, :ci] = w
            self.bias = torch.zeros(pco, device=w.device, dtype=torch.float32)
            self.bias[:co] = b

    def unpad(self, grads, out):
        """move this layer's gradients (keyed by the padded tensors) onto the real parameters."""
        co, ci = self.real.weight.shape[0], self.real.weight.shape[1]
        if self.weight in grads:
            out[self.real.weight] = grads[self.weight][:co, :ci].contiguous()
            out[self.real.bias] = grads[self.bias][:co].contiguous()


class _PaddedBlock:
    def __init__(self, blk):
        self.c1, self.c2 = _PaddedConv(blk.c1), _PaddedConv(blk.c2)


class DiscriminatorFunction(torch.autograd.Function):
    """``Discriminator.calclate_logit`` (``real is None``) and ``calclate_logit_and_feature_matching`` (vae.py:149-171).
    With a real batch the two batches run as ONE batch of 2B samples (every layer is per-sample), the fake half feeding the logit
    and both halves the per-stage L1 feature distance.  Returns (logit, feat_loss); feat_loss is a zero scalar without ``real``."""

    @staticmethod
    def forward(fctx, disc, fake, real, *params):
        from torch import nn
        dev = fake.device
        nf = fake.shape[0]
        x = fake.contiguous().float() if real is None else torch.cat([fake.float(), real.float()], 0).contiguous()
        b, cin, h, w = x.shape
        c0 = disc.input_layer.weight.shape[0]
        p0 = _pad32(c0)                                                       # the stem writes the padded width directly
        w_in = torch.zeros(p0, cin, device=dev, dtype=torch.float32)
        w_in[:c0] = w2d(disc.input_layer)
        b_in = torch.zeros(p0, device=dev, dtype=torch.float32)
        b_in[:c0] = disc.input_layer.bias.detach()
        rows = torch.empty(b * h * w, p0, device=dev, dtype=torch.float32)
        ops.stem_nchw(x, w_in, b_in, rows, b, cin, h * w, p0)
        logit = torch.zeros(1, device=dev, dtype=torch.float32)
        feat = torch.zeros(1, device=dev, dtype=torch.float32)
        tape = []
        for stage, down, exit_conv in zip(disc.stages, disc.downsamples, disc.early_exits):
            rec = dict(blocks=[], pblocks=[_PaddedBlock(blk) for blk in stage.seq], shape=(b, h, w), down=None)
            for pblk in rec["pblocks"]:
                rows = _resblock_forward(pblk, rows, (b, h, w), rec["blocks"])
            m, cp = rows.shape
            c = exit_conv.weight.shape[1]
            mf = nf * h * w                                                    # rows of the fake half
            if real is not None:                                              # (fake_x - real_x).abs().mean() over the c real channels
                part = torch.empty(1, device=dev, dtype=torch.float32)
                ops.l1_loss(rows[:mf], rows[mf:], part)
                feat.add_(part, alpha=cp / c)                                   # scalar bookkeeping only (the padded zeros add 0 to the sum)
            # c(fake_x).mean(): the 1-channel 1x1 conv through the NHWC -> NCHW head kernel, then one column sum
            w_exit = torch.zeros(cp, 1, device=dev, dtype=torch.float32)
            w_exit[:c, 0] = exit_conv.weight.detach().reshape(-1)
            e = torch.empty(nf, 1, h, w, device=dev, dtype=torch.float32)
            ops.head_nchw(rows[:mf], w_exit, exit_conv.bias.detach(), e, nf, cp, h * w, 1)
            logit.add_(ops.colsum(e, mf, 1), alpha=1.0 / mf)
            rec.update(rows=rows, exit=exit_conv, w_exit=w_exit, c=c)
            if not isinstance(down, nn.Identity):                              # Conv2d(c, c', 2, 2): space-to-depth + GEMM
                pd = _PaddedConv(down)
                pco, pci = pd.weight.shape[0], pd.weight.shape[1]
                packed = pd.weight.permute(0, 2, 3, 1).reshape(pco, 4 * pci).contiguous()           # [co][(dy, dx, ci)]
                s2d = ops.space_to_depth2(rows, b, h // 2, w // 2, cp)
                h, w = h // 2, w // 2
                rows = torch.empty(b * h * w, pco, device=dev, dtype=torch.float32)
                ops.gemm(s2d, b * h * w, pco, 4 * pci, [packed], rows, biases=[pd.bias])
                rec["down"] = dict(pd=pd, packed=packed, s2d=s2d)
            tape.append(rec)
        fctx.disc, fctx.tape, fctx.x, fctx.params, fctx.nf, fctx.has_real, fctx.w_in = disc, tape, x, params, nf, real is not None, w_in
        return logit.reshape(()), feat.reshape(())

    @staticmethod
    def backward(fctx, glogit, gfeat):
        disc, tape, x, nf = fctx.disc, fctx.tape, fctx.x, fctx.nf
        dev = x.device
        pg, grads = {}, {}                              # gradients keyed by padded stand-ins / by the real parameters
        glogit = glogit.reshape(1).contiguous().float()
        gfeat = gfeat.reshape(1).contiguous().float()
        drows = None
        for rec in reversed(tape):
            b, h, w = rec["shape"]
            rows, cp, c = rec["rows"], rec["rows"].shape[1], rec["c"]
            m, mf = rows.shape[0], nf * h * w
            if rec["down"] is not None:
                pd, packed, s2d = rec["down"]["pd"], rec["down"]["packed"], rec["down"]["s2d"]
                pco, pci = pd.weight.shape[0], pd.weight.shape[1]
                dr = _Rows(drows)
                dwp = grad_weight_rows(dr, _Rows(s2d), drows.shape[0])                          # [co][(dy, dx, ci)]
                pg[pd.weight] = dwp.reshape(pco, 2, 2, pci).permute(0, 3, 1, 2).contiguous()
                pg[pd.bias] = dr.colsum().clone()
                pd.unpad(pg, grads)
                fine = torch.empty(m, pci, device=dev, dtype=torch.float32)
                ops.gemm(drows, drows.shape[0], 4 * pci, pco, [packed.t().contiguous()], fine, ldo=pci, o_mode=ops.O_CONVT2X2,
                         out_hw=(h // 2, w // 2), cout=pci)                                   # depth-to-space in the epilogue
                drows = fine
            else:
                drows = torch.zeros(m, cp, device=dev, dtype=torch.float32)
            # early exit: logit += mean_p (rows[p] . w + b) over the fake half
            dout = (glogit * (1.0 / mf)).expand(mf).contiguous().reshape(nf, 1, h, w)
            dexit = torch.empty(mf, cp, device=dev, dtype=torch.float32)
            dw_e = torch.empty(cp, 1, device=dev, dtype=torch.float32)
            db_e = torch.empty(1, device=dev, dtype=torch.float32)
            ops.head_bwd(rows[:mf], rec["w_exit"], dout, dexit, dw_e, db_e, nf, cp, h * w, 1)
            ops.add_(drows[:mf], dexit)
            grads[rec["exit"].weight] = dw_e[:c, 0].reshape(rec["exit"].weight.shape).contiguous()
            grads[rec["exit"].bias] = db_e
            if fctx.has_real:                                                 # d mean|fake - real| (x cp / c: the mean is over c channels)
                gs = gfeat * (cp / c)
                dl = torch.empty(mf, cp, device=dev, dtype=torch.float32)
                ops.l1_loss_bwd(rows[:mf], rows[mf:], gs, dl)
                ops.add_(drows[:mf], dl)
                ops.l1_loss_bwd(rows[mf:], rows[:mf], gs, dl)
                ops.add_(drows[mf:], dl)
            for blk_rec, pblk in zip(reversed(rec["blocks"]), reversed(rec["pblocks"])):
                drows = _resblock_backward(blk_rec, drows, (b, h, w), pg)
                pblk.c1.unpad(pg, grads)
                pblk.c2.unpad(pg, grads)
        b, cin, h0, w0 = x.shape
        c0 = disc.input_layer.weight.shape[0]
        p0 = drows.shape[1]
        dw0 = torch.empty(p0, cin, device=dev, dtype=torch.float32)
        ops.stem_bwd(x, drows, dw0, b, cin, h0 * w0, p0)
        grads[disc.input_layer.weight] = dw0[:c0].reshape(disc.input_layer.weight.shape).contiguous()
        grads[disc.input_layer.bias] = ops.colsum(drows, drows.shape[0], p0)[:c0].contiguous()
        dfake = None
        if fctx.needs_input_grad[1]:                                          # real_x.requires_grad = False (vae.py:150)
            dfake = torch.empty(nf, cin, h0, w0, device=dev, dtype=torch.float32)
            ops.head_nchw(drows[:nf * h0 * w0], fctx.w_in, None, dfake, nf, p0, h0 * w0, cin)
        return (None, dfake, None) + tuple(grads.get(p) for p in fctx.params)
