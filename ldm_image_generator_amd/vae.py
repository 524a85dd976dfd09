"""VAE decoder path (reference: vae.py:45-66,99-132), MI355X-native host side.

``Decoder`` / ``DecoderStack`` / ``ResBlock`` / ``VAE.decode`` keep the reference's
constructor arguments and its 50 ``state_dict`` keys (incl. the dead
``output_layer``).  Every convolution is one launch of the fp32-MFMA GEMM family:
dense 3x3 as implicit GEMM with fused bias + leaky_relu (+ residual),
ConvTranspose2d(2,2) as one GEMM with a 2x2 scatter epilogue; ``to_rgb`` and the
bilinear RGB accumulation are one fused HBM-bound kernel per stage.

``Encoder`` (latent pre-encoding for train_ldm.py) reuses the same kernels.  SURVEY 8f.4 (VAE training) is started
with its integer part: ``VectorQuantizer`` (quantize with the reference's indices, embed, the two-sided L1 loss with gradients) and
``VAE.calclate_loss``; ``Encoder`` and ``Decoder`` are differentiable (``vae_train.py``: parameter gradients and input gradients,
pinned against the reference's autograd), so the VAE objective trains end to end.  ``Discriminator`` (train_vae.py's adversarial
term) runs forward and backward on the same kernels, its 48-channel stages carried at 64 zero-padded columns.
"""
import torch
import torch.nn as nn

from . import ops, weights
from .modules import from_rows, to_rows, w2d


class _PackedWeight:
    """Cache of a re-laid-out weight, invalidated by in-place updates / device moves."""

    def __init__(self, fn):
        self.fn = fn
        self.key = None
        self.val = None

    def get(self, w):
        key = weights.key(w)
        if key != self.key:
            self.key, self.val = key, self.fn(w.detach()).contiguous()
        return self.val


def _pack3x3(w):       # [Cout, Cin, 3, 3] -> [Cout][tap][Cin]
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], 9 * w.shape[1])


def _pack_convt(w):    # ConvTranspose2d [Cin, Cout, 2, 2] -> [(dy, dx, co)][ci]
    return w.permute(2, 3, 1, 0).reshape(4 * w.shape[1], w.shape[0])


def _bf16(fn):
    """A weight re-layout followed by one RNE rounding to bf16 (ldm_cast_bf16), for _PackedWeight."""
    return lambda w: ops.cast_bf16(fn(w).contiguous())


def conv3x3_rows16(rows16, shape, conv, packed16, slope=0.01, addend16=None):
    """conv3x3_rows in the bf16 decode mode: bf16 rows in, bf16 rows out, fp32 accumulate; the skip is added in fp32 before the
    one rounding of the result."""
    b, h, w = shape
    cin, cout = conv.weight.shape[1], conv.weight.shape[0]
    out = torch.empty(rows16.shape[0], cout, device=rows16.device, dtype=torch.bfloat16)
    ops.gemm_bf16(rows16, rows16.shape[0], cout, 9 * cin, [packed16.get(conv.weight)], out, ldw=9 * cin, biases=[conv.bias.detach()],
                  act=ops.ACT_LRELU, slope=slope, addend=addend16, a_mode=ops.A_CONV3X3, conv_hw=(h, w), cin=cin)
    return out


def conv3x3_rows(rows, shape, conv, packed, slope=0.01, addend=None):
    """lrelu(conv3x3(x) + bias) (+ addend) on channels-last rows."""
    b, h, w = shape
    cin, cout = conv.weight.shape[1], conv.weight.shape[0]
    out = torch.empty(rows.shape[0], cout, device=rows.device, dtype=torch.float32)
    ops.gemm(rows, rows.shape[0], cout, 9 * cin, [packed.get(conv.weight)], out, lda=cin, ldw=9 * cin,
             biases=[conv.bias.detach()], act=ops.ACT_LRELU, slope=slope, addend=addend,
             a_mode=ops.A_CONV3X3, conv_hw=(h, w), cin=cin)
    return out


class ResBlock(nn.Module):
    """x + lrelu(c2(lrelu(c1(x))))  (vae.py:54-66)."""

    def __init__(self, channels):
        super().__init__()
        self.c1 = nn.Conv2d(channels, channels, 3, 1, 1)
        self.c2 = nn.Conv2d(channels, channels, 3, 1, 1)
        self._p1 = _PackedWeight(_pack3x3)
        self._p2 = _PackedWeight(_pack3x3)
        self._p1h = _PackedWeight(_bf16(_pack3x3))
        self._p2h = _PackedWeight(_bf16(_pack3x3))

    def forward_rows(self, rows, shape):
        if rows.dtype == torch.bfloat16:                      # bf16 decode mode (autocast.set_compute_dtype)
            y = conv3x3_rows16(rows, shape, self.c1, self._p1h)
            return conv3x3_rows16(y, shape, self.c2, self._p2h, addend16=rows)
        y = conv3x3_rows(rows, shape, self.c1, self._p1)
        return conv3x3_rows(y, shape, self.c2, self._p2, addend=rows)

    def forward(self, x):
        rows, shape = to_rows(x)
        return from_rows(self.forward_rows(rows, shape), shape)


class ResStack(nn.Module):
    """vae.py:68-74."""

    def __init__(self, channels, num_layers=2):
        super().__init__()
        self.seq = nn.Sequential(*[ResBlock(channels) for _ in range(num_layers)])

    def forward_rows(self, rows, shape):
        for blk in self.seq:
            rows = blk.forward_rows(rows, shape)
        return rows

    def forward(self, x):
        rows, shape = to_rows(x)
        return from_rows(self.forward_rows(rows, shape), shape)


class Encoder(nn.Module):
    """vae.py:76-96: image -> latent (used by train_ldm.py to pre-encode the training set).
    Same kernels as the decoder: stem 1x1 from NCHW, dense 3x3 implicit GEMMs, avg-pool + 1x1 GEMM, and the
    NHWC->NCHW head kernel for the 8-channel output."""

    def __init__(self, input_channels=3, latent_channels=8, channels=[64, 128, 256, 512], stages=[2, 2, 2, 2]):
        super().__init__()
        self.input_layer = nn.Conv2d(input_channels, channels[0], 1, 1, 0)
        self.output_layer = nn.Conv2d(channels[-1], latent_channels, 1, 1, 0)
        self.stages = nn.ModuleList([ResStack(c, l) for c, l in zip(channels, stages)])
        self.downsamples = nn.ModuleList([])
        for i, c in enumerate(channels):
            if i == len(self.stages) - 1:
                self.downsamples.append(nn.Identity())
            else:
                self.downsamples.append(nn.Sequential(nn.AvgPool2d(kernel_size=2), nn.Conv2d(c, channels[i + 1], 1, 1, 0)))
        self._out_t = _PackedWeight(lambda w: w.reshape(w.shape[0], -1).t())          # [C, latent] for the head kernel

    def forward(self, x):
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from .vae_train import EncoderFunction                     # training: same kernels + tape, hand-written backward
            return EncoderFunction.apply(self, x, *[p for p in self.parameters() if p.requires_grad])
        b, cin, h, w = x.shape
        dev = x.device
        c0 = self.input_layer.weight.shape[0]
        rows = torch.empty(b * h * w, c0, device=dev, dtype=torch.float32)
        ops.stem_nchw(x.contiguous().float(), w2d(self.input_layer), self.input_layer.bias.detach(), rows, b, cin, h * w, c0)
        for stage, down in zip(self.stages, self.downsamples):
            rows = stage.forward_rows(rows, (b, h, w))
            if not isinstance(down, nn.Identity):
                conv = down[1]
                c = rows.shape[1]
                pooled = torch.empty(b * (h // 2) * (w // 2), c, device=dev, dtype=torch.float32)
                ops.avgpool2(rows, pooled, b, h, w, c)
                h, w = h // 2, w // 2
                rows = torch.empty(b * h * w, conv.weight.shape[0], device=dev, dtype=torch.float32)
                ops.gemm(pooled, b * h * w, conv.weight.shape[0], c, [w2d(conv)], rows, biases=[conv.bias.detach()])
        cz = self.output_layer.weight.shape[0]
        z = torch.empty(b, cz, h, w, device=dev, dtype=torch.float32)
        ops.head_nchw(rows, self._out_t.get(self.output_layer.weight), self.output_layer.bias.detach(), z, b, rows.shape[1], h * w, cz)
        return z


class DecoderStack(nn.Module):
    def __init__(self, channels, num_layers, output_channels=3):
        super().__init__()
        if not 1 <= output_channels <= 4:
            raise NotImplementedError("the fused to_rgb kernel handles 1 to 4 output channels (got %r)" % (output_channels,))
        self.layers = nn.Sequential(*[ResBlock(channels) for _ in range(num_layers)])
        self.to_rgb = nn.Conv2d(channels, output_channels, 1, 1, 0)

    def forward_rows(self, rows, shape, prev_rgb):
        for blk in self.layers:
            rows = blk.forward_rows(rows, shape)
        b, h, w = shape
        rgb = torch.empty(b, self.to_rgb.weight.shape[0], h, w, device=rows.device, dtype=torch.float32)
        head = ops.rgb_head_bf16 if rows.dtype == torch.bfloat16 else ops.rgb_head
        head(rows, w2d(self.to_rgb), self.to_rgb.bias.detach(), prev_rgb, rgb, b, h, w, rows.shape[1])
        return rows, rgb

    def forward(self, x):
        rows, shape = to_rows(x)
        rows, rgb = self.forward_rows(rows, shape, None)
        return from_rows(rows, shape), rgb


class Decoder(nn.Module):
    def __init__(self, output_channels=3, latent_channels=8, channels=[512, 256, 128, 64], stages=[2, 2, 2, 2]):
        super().__init__()
        self.input_layer = nn.Conv2d(latent_channels, channels[0], 1, 1, 0)
        self.output_layer = nn.Conv2d(channels[-1], output_channels, 1, 1, 0)       # dead weight in the reference too
        self.stages = nn.ModuleList([DecoderStack(c, l, output_channels=output_channels) for c, l in zip(channels, stages)])
        self.upsamples = nn.ModuleList([])
        for i, c in enumerate(channels):
            if i == 0:
                self.upsamples.append(nn.Identity())
            else:
                self.upsamples.append(nn.ConvTranspose2d(channels[i - 1], c, 2, 2, 0))
        self._up_packed = [_PackedWeight(_pack_convt) for _ in channels]
        self._up_packed16 = [_PackedWeight(_bf16(_pack_convt)) for _ in channels]
        self._up_bias4 = [_PackedWeight(lambda b: b.repeat(4)) for _ in channels]       # bias per (dy, dx, co) column of the 2x2 GEMM
        # None: exact fp32 (default).  torch.bfloat16 (autocast.set_compute_dtype): activations travel as bf16 rows, every conv
        # accumulates in fp32 on v_mfma_f32_32x32x16_bf16, RGB planes stay fp32.  An extension beyond the reference, which decodes
        # in fp32 outside its autocast region (sample_ldm.py:73-74).
        self.compute_dtype = None

    def forward(self, x):
        """vae.py:122-132: z [B, latent, h, w] -> RGB [B, 3, 8h, 8w] (NCHW)."""
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            # training: same kernels + tape, hand-written backward (vae_train.py); wrap pure inference in torch.no_grad()
            from .vae_train import DecoderFunction
            return DecoderFunction.apply(self, x, *[p for p in self.parameters() if p.requires_grad])
        b, cz, h, w = x.shape
        dev = x.device
        c0 = self.input_layer.weight.shape[0]
        if self.compute_dtype is torch.bfloat16:
            return self._forward_bf16(x)
        rows = torch.empty(b * h * w, c0, device=dev, dtype=torch.float32)
        ops.stem_nchw(x.contiguous().float(), w2d(self.input_layer), self.input_layer.bias.detach(), rows, b, cz, h * w, c0)
        rgb = None
        for i, (up, stage) in enumerate(zip(self.upsamples, self.stages)):
            if not isinstance(up, nn.Identity):
                cin, cout = up.weight.shape[0], up.weight.shape[1]
                fine = torch.empty(b * 4 * h * w, cout, device=dev, dtype=torch.float32)
                ops.gemm(rows, b * h * w, 4 * cout, cin, [self._up_packed[i].get(up.weight)], fine,
                         biases=[up.bias.detach()], ldo=cout, o_mode=ops.O_CONVT2X2, out_hw=(h, w), cout=cout)
                rows = fine
                h, w = 2 * h, 2 * w
            rows, rgb = stage.forward_rows(rows, (b, h, w), rgb)
        return rgb


def _decoder_forward_bf16(self, x):
    """Decoder.forward with bf16 activations (vae.py:122-132 under a 16-bit compute type): the input layer writes bf16 rows, every
    dense 3x3 is a bf16 implicit GEMM (fp32 accumulate, bias + leaky_relu + skip in fp32, one rounding), ConvTranspose2d(2, 2) is
    a bf16 GEMM to (dy, dx, co) columns followed by the depth-to-space pass, to_rgb + bilinear accumulation stay fp32 planes."""
    b, cz, h, w = x.shape
    dev = x.device
    chans = [self.input_layer.weight.shape[0]] + [up.weight.shape[1] for up in self.upsamples if not isinstance(up, nn.Identity)]
    if any(c % 64 for c in chans):
        raise NotImplementedError("bf16 decode needs every stage width to be a multiple of 64 (got %r)" % (chans,))
    c0 = chans[0]
    rows = torch.empty(b * h * w, c0, device=dev, dtype=torch.bfloat16)
    ops.stem_nchw_bf16(x.contiguous().float(), w2d(self.input_layer), self.input_layer.bias.detach(), rows, b, cz, h * w, c0)
    rgb = None
    for i, (up, stage) in enumerate(zip(self.upsamples, self.stages)):
        if not isinstance(up, nn.Identity):
            cin, cout = up.weight.shape[0], up.weight.shape[1]
            quad = torch.empty(b * h * w, 4 * cout, device=dev, dtype=torch.bfloat16)
            ops.gemm_bf16(rows, b * h * w, 4 * cout, cin, [self._up_packed16[i].get(up.weight)], quad, biases=[self._up_bias4[i].get(up.bias)])
            rows = ops.depth_to_space2_bf16(quad, b, h, w, cout)
            del quad
            h, w = 2 * h, 2 * w
        rows, rgb = stage.forward_rows(rows, (b, h, w), rgb)
    return rgb


Decoder._forward_bf16 = _decoder_forward_bf16


class _VQLossFunction(torch.autograd.Function):
    """vae.py:12-16: l1(x, e.detach()) + l1(e, x.detach()) with e = embeddings[quantize(x)] -- two kernels each way."""

    @staticmethod
    def forward(fctx, x_rows, emb):
        x_rows, emb = x_rows.contiguous().float(), emb.contiguous().float()
        idx = ops.vq_quantize(x_rows, emb)
        e_rows = ops.vq_embed(idx, emb)
        fctx.save_for_backward(x_rows, e_rows, idx)
        fctx.n_emb = emb.shape[0]
        return ops.vq_loss(x_rows, e_rows).reshape(())

    @staticmethod
    def backward(fctx, gout):
        x_rows, e_rows, idx = fctx.saved_tensors
        dx, demb = ops.vq_loss_bwd(x_rows, e_rows, idx, gout.reshape(1).contiguous().float(), fctx.n_emb)
        return dx, demb


class VectorQuantizer(nn.Module):
    """vae.py:7-26.  ``quantize`` returns the reference's indices: nearest codebook row under torch.cdist's own rounding, first
    index on ties (ldm_vq_quantize_f32)."""

    def __init__(self, num_embeddings=8192, dim=8):
        super().__init__()
        self.embeddings = nn.Parameter(torch.randn(num_embeddings, dim))

    def _rows(self, x):
        if x.shape[-1] != self.embeddings.shape[1]:
            raise ValueError("last dimension of x must be the embedding dimension %d" % self.embeddings.shape[1])
        return x.reshape(-1, x.shape[-1]).contiguous().float()

    def calculate_loss(self, x):
        return _VQLossFunction.apply(self._rows(x), self.embeddings)

    @torch.no_grad()
    def quantize(self, x):
        return ops.vq_quantize(self._rows(x), self.embeddings.detach().contiguous()).reshape(x.shape[:-1])

    def embed(self, x):
        """vae.py:24-26 (F.embedding): indices -> codebook rows.  Like F.embedding an out-of-range index raises (checked on the
        device, one host sync; this is not on the sampling hot path).  The gather is not differentiable here: the codebook's
        gradient flows through ``calculate_loss`` only, which is the only place the reference's training loop uses it."""
        idx = x.reshape(-1).contiguous()
        n = self.embeddings.shape[0]
        if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= n):
            raise IndexError("VectorQuantizer.embed: index out of range [0, %d)" % n)
        out = ops.vq_embed(idx, self.embeddings.detach().contiguous())
        return out.reshape(tuple(x.shape) + (self.embeddings.shape[1],))


class VAE(nn.Module):
    """vae.py:30-52.  ``encode`` / ``decode`` are the sampling path; ``calclate_loss`` (sic) is the VAE training
    objective, differentiable end to end (``vae_train.py``)."""

    def __init__(self, encoder, decoder, quantizer):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.quantizer = quantizer

    def calclate_loss(self, x, noise_gain=0.1):
        """-> (loss_recon, loss_reg, y) as vae.py:36-43; differentiable: Encoder, Decoder and the quantizer's embeddings get their
        gradients through hand-written backward kernels (vae_train.py, train.L1LossFunction, _VQLossFunction)."""
        z = self.encoder(x)
        noise = torch.randn(z.shape, device=x.device)
        if torch.is_grad_enabled() and z.requires_grad:
            from .train import L1LossFunction
            from .vae_train import AddNoiseFunction, ToRowsFunction
            z = AddNoiseFunction.apply(z, noise, float(noise_gain))
            loss_reg = self.quantizer.calculate_loss(ToRowsFunction.apply(z))
            y = self.decoder(z)
            return L1LossFunction.apply(y, x.detach()), loss_reg, y
        b = z.shape[0]
        zn = torch.empty_like(z)                          # z * 1 + noise * gain, products and sum rounded like torch's two ops
        ops.qsample(z, noise, torch.ones(b, device=x.device), torch.full((b,), float(noise_gain), device=x.device), zn)
        z = zn
        rows, _ = to_rows(z)                              # [B*h*w, 8] == z.reshape(B, C, -1).transpose(1, 2)
        loss_reg = self.quantizer.calculate_loss(rows)
        y = self.decoder(z)
        loss = torch.empty(1, device=x.device, dtype=torch.float32)
        ops.l1_loss(x.contiguous().float(), y, loss)
        return loss.reshape(()), loss_reg, y

    @torch.no_grad()
    def encode(self, x):
        return self.encoder(x)

    @torch.no_grad()
    def decode(self, z):
        return self.decoder(z)


class Discriminator(nn.Module):
    """vae.py:134-171: the multi-scale critic of train_vae.py (adversarial term + its own hinge loss).  Same constructor, parameter
    names and registration order as the reference (``input_layer``, ``stages.*``, ``downsamples.*``, ``early_exits.*``); both
    methods are differentiable w.r.t. the parameters and the fake batch (``vae_train.DiscriminatorFunction``)."""

    def __init__(self, input_channels=3, channels=[32, 48, 48, 96], stages=[2, 2, 2, 2], stem_size=1):
        super().__init__()
        if not (isinstance(stem_size, int) and stem_size >= 1):
            raise ValueError("stem_size must be a positive integer")
        self.stem_size = stem_size          # > 1: a stride-s patchify conv = 1x1 over the pixel_unshuffle'd image (see unet.UNet)
        self.input_layer = nn.Conv2d(input_channels, channels[0], stem_size, stem_size, 0)
        self.stages = nn.ModuleList([ResStack(c, l) for c, l in zip(channels, stages)])
        self.early_exits = nn.ModuleList([])
        self.downsamples = nn.ModuleList([])
        for i, c in enumerate(channels):
            if i == len(self.stages) - 1:
                self.downsamples.append(nn.Identity())
            else:
                self.downsamples.append(nn.Conv2d(c, channels[i + 1], 2, 2, 0))
            self.early_exits.append(nn.Conv2d(c, 1, 1, 1, 0))

    def _run(self, fake_x, real_x):
        from .vae_train import DiscriminatorFunction
        if self.stem_size > 1:
            import torch.nn.functional as F
            fake_x = F.pixel_unshuffle(fake_x, self.stem_size)
            real_x = None if real_x is None else F.pixel_unshuffle(real_x, self.stem_size)
        return DiscriminatorFunction.apply(self, fake_x, real_x, *[p for p in self.parameters() if p.requires_grad])

    def calclate_logit_and_feature_matching(self, fake_x, real_x):
        real_x.requires_grad = False                      # as the reference (vae.py:150)
        return self._run(fake_x, real_x)

    def calclate_logit(self, fake_x):
        return self._run(fake_x, None)[0]


def to_uint8_images(img):
    """sample_ldm.py:75-77 on the device: [B,3,H,W] float -> [B,H,W,3] uint8 (truncation)."""
    b, c, h, w = img.shape
    out = torch.empty(b, h, w, c, device=img.device, dtype=torch.uint8)
    ops.to_uint8_hwc(img.contiguous(), out, b, c, h * w)
    return out
