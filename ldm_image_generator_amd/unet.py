"""Swin-style UNet (reference: unet.py:9-103), MI355X-native host side.

Drop-in surface: ``UNet(input_channels=8, stages=[3,3,9,3], channels=[128,256,512,1024],
stem_size=1).forward(x, time, condition=None)`` with NCHW tensors, the reference's
``state_dict`` keys (1376 tensors at the default size), the reference's default
initialisation (same module construction order, so the same ``torch.manual_seed``
gives the same weights) and the reference's consumption of Python's global
``random`` (stochastic depth + expert choice, in block order).

Inside, activations are channels-last rows [M = B*H*W, C]; per SwinBlock the
launches are
    channelnorm_film -> grouped-3x3 GEMM (+bias +residual)
                     -> [in-proj GEMM -> window attention -> out-proj GEMM (+=)]
                     -> gated GEMM (a*relu(b), 3 experts by pointer) -> GEMM over stacked K (+=)
and the FiLM (mul | bias) of each block is produced once per distinct timestep
by two GEMMs over the sin/cos code table (batch-independent).
"""
import random

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, sinusoidal, weights
from .attention import CrossAttention, WindowAttention
from .modules import ChannelNorm, RandomMoE, from_rows, to_rows, w2d
from .sinusoidal import PositionalEncoding2d, TimeEncoding2d


# bumped on every nn.Module.register_parameter in the process: a cheap "some Parameter object may have been replaced" signal
_PARAM_GENERATION = [0]


def _bump_param_generation(module, name, param):
    _PARAM_GENERATION[0] += 1


torch.nn.modules.module.register_module_parameter_registration_hook(_bump_param_generation)


class TimeContext:
    """Per-forward bookkeeping: distinct timesteps, sample->slot map, and the
    sin/cos code table of every (C, H, W) seen so far (shared by all blocks of a level)."""

    def __init__(self, time, batch, device, uniform=None, dedupe=True):
        self.unique_slots = False
        if not dedupe:
            # training: one slot per sample, no dedupe and no host sync -- the FiLM gradient of a (slot, pixel) row then
            # has exactly one writer (plain stores instead of atomics); samples that happen to share a timestep simply
            # carry identical code rows, and their gradients meet in the weight-gradient sums
            self.t_unique = time.detach().to(device=device, dtype=torch.int64)
            self.slot = torch.arange(batch, dtype=torch.int32, device=device)
            self.unique_slots = True
        elif uniform is not None:
            # (value, 1-element int64 device tensor): no host->device copy (= stream sync) inside the denoise loop
            self.t_unique = uniform[1] if isinstance(uniform, tuple) else torch.tensor([int(uniform)], dtype=torch.int64, device=device)
            self.slot = None
        else:
            tc = time.detach().to("cpu", torch.int64)             # one host sync per forward
            uniq, inv = torch.unique(tc, return_inverse=True)
            self.t_unique = uniq.to(device)
            self.slot = None if uniq.numel() == 1 else inv.to(device=device, dtype=torch.int32)
        self.batch = batch
        self._codes = {}

    def codes(self, channels, height, width):
        key = (channels, height, width)
        if key not in self._codes:
            self._codes[key] = sinusoidal.embed(self.t_unique, height, width, channels)
        return self._codes[key]


class Encodings(nn.Module):
    """FiLM from position + time codes (unet.py:9-23)."""

    def __init__(self, channels):
        super().__init__()
        self.proj1 = nn.Conv2d(channels * 2, channels * 4, 1, 1, 0)
        self.act = nn.ReLU()
        self.proj2 = nn.Conv2d(channels * 4, channels * 2, 1, 1, 0)
        self.pe = PositionalEncoding2d(channels, return_encoding_only=True)
        self.te = TimeEncoding2d(channels, return_encoding_only=True)
        self.channels = channels

    def film_rows(self, codes):
        """codes [nT*HW, 2C] -> (mul | bias) rows [nT*HW, 2C]  (unet.py:20)."""
        m = codes.shape[0]
        c = self.channels
        hid = torch.empty(m, 4 * c, device=codes.device, dtype=torch.float32)
        ops.gemm(codes, m, 4 * c, 2 * c, [w2d(self.proj1)], hid, biases=[self.proj1.bias.detach()], act=ops.ACT_RELU)
        film = torch.empty(m, 2 * c, device=codes.device, dtype=torch.float32)
        ops.gemm(hid, m, 2 * c, 4 * c, [w2d(self.proj2)], film, biases=[self.proj2.bias.detach()])
        return film

    def forward(self, x, t):
        rows, (b, h, w) = to_rows(x)
        ctx = TimeContext(t, b, x.device)
        film = self.film_rows(ctx.codes(self.channels, h, w))
        out = torch.empty_like(rows)
        ops.film(rows, film, ctx.slot, out, b, h * w, self.channels)
        return from_rows(out, (b, h, w))


class SwinBlock(nn.Module):
    def __init__(self, channels, head_dim=32, window_size=6, shift=0, attention=True, stochastic_depth=0.25):
        super().__init__()
        if head_dim != 32:
            raise NotImplementedError("HIP kernels are specialised for head_dim == 32 (the reference's only value)")
        self.norm = ChannelNorm(channels)
        self.ffn = RandomMoE(channels)
        self.conv = nn.Conv2d(channels, channels, 3, 1, 1, groups=channels // head_dim)
        self.stochastic_depth = stochastic_depth
        self.attention_flag = attention
        if attention:
            self.self_attention = WindowAttention(channels, n_heads=channels // head_dim, window_size=window_size, shift=shift)
            self.cross_attention = CrossAttention(channels, n_heads=channels // head_dim)
        self.encodings = Encodings(channels)
        self.channels = channels
        self._packed = None

    def _conv_weight(self):
        """[C, 32, 3, 3] -> [C][tap][32] (K contiguous for the implicit-GEMM), cached per weight version."""
        w = self.conv.weight
        key = weights.key(w)
        if self._packed is None or self._packed[0] != key:
            c = w.shape[0]
            self._packed = (key, w.detach().permute(0, 2, 3, 1).reshape(c, 9 * w.shape[1]).contiguous())
        return self._packed[1]

    def draw(self):
        """The block's Python-RNG decisions, in the reference's order (unet.py:39, modules.py:35):
        returns None when stochastic depth skips the block, else the two expert indices."""
        if self.training and random.random() <= self.stochastic_depth:
            return None
        return self.ffn.pick()

    def forward_rows(self, rows, shape, ctx, decision=False, film=None):
        """unet.py:38-48 on channels-last rows.  ``decision``/``film`` may be supplied by UNet.forward,
        which draws all decisions up front (same RNG order) and batches every block's FiLM MLP."""
        picks = self.draw() if decision is False else decision
        if picks is None:
            return rows
        b, h, w = shape
        m, c = rows.shape
        if film is None:
            film = self.encodings.film_rows(ctx.codes(c, h, w))
        xf = torch.empty_like(rows)
        ops.channelnorm_film(rows, film, ctx.slot, xf, b, h * w, c, self.norm.eps)
        # y = conv(xf) + bias + res
        y = torch.empty_like(rows)
        g = c // 32
        ops.gemm(xf, m, 32, 288, [self._conv_weight()], y, lda=c, ldw=288, biases=[self.conv.bias.detach()],
                 addend=rows, ldadd=c, ldo=c, a_mode=ops.A_CONV3X3, conv_hw=(h, w), cin=32,
                 groups=g, a_gstride=32, w_gstride=32 * 288, o_gstride=32, b_gstride=32)
        if self.attention_flag:
            self.self_attention.forward_rows(xf, shape, addend=y, out=y)
        self.ffn.forward_rows(xf, addend=y, out=y, picks=picks)
        return y

    def forward(self, x, t, c=None):
        if c is not None and self.attention_flag:
            raise NotImplementedError("cross-attention has no defined semantics in the reference (attention.py:92-98)")
        rows, shape = to_rows(x)
        out = self.forward_rows(rows, shape, TimeContext(t, shape[0], x.device))
        return from_rows(out, shape)


class SwinStack(nn.Module):
    def __init__(self, channels, head_dim=32, window_size=6, num_blocks=2, attention=True):
        super().__init__()
        self.blocks = nn.ModuleList([])
        for i in range(num_blocks):
            shift = window_size // 2 if i % 2 == 0 else 0
            flag_attn = attention if i >= num_blocks - 2 else False       # unet.py:57
            self.blocks.append(SwinBlock(channels, head_dim, window_size, shift, attention=flag_attn))

    def forward_rows(self, rows, shape, ctx, plan=None):
        for blk in self.blocks:
            if plan is None:
                rows = blk.forward_rows(rows, shape, ctx)
            else:
                rows = blk.forward_rows(rows, shape, ctx, *plan[blk])
        return rows

    def forward(self, x, t, c=None):
        rows, shape = to_rows(x)
        return from_rows(self.forward_rows(rows, shape, TimeContext(t, shape[0], x.device)), shape)


class UNetBlock(nn.Module):
    def __init__(self, stage, ch_conv):
        super().__init__()
        self.stage = stage
        self.ch_conv = ch_conv


class UNet(nn.Module):
    def __init__(self, input_channels=8, stages=[3, 3, 9, 3], channels=[128, 256, 512, 1024], stem_size=1):
        super().__init__()
        if not (isinstance(stem_size, int) and stem_size >= 1):
            raise ValueError("stem_size must be a positive integer")
        # stem_size = s > 1 (unet.py:77-78: a stride-s patchify conv in, its ConvTranspose2d out): forward() re-tiles the image with
        # pixel_unshuffle / pixel_shuffle and every path below sees a 1x1 stem / head over input_channels * s * s channels -- the two
        # weights [C0, Cin, s, s] ARE the [C0, Cin s^2] matrices of that 1x1 problem, in place
        self.stem_size = stem_size
        self.encoder_first = nn.Conv2d(input_channels, channels[0], stem_size, stem_size, 0)
        self.decoder_last = nn.ConvTranspose2d(channels[0], input_channels, stem_size, stem_size, 0)
        self.encoder_stages = nn.ModuleList([])
        self.decoder_stages = nn.ModuleList([])
        for i, (l, c) in enumerate(zip(stages, channels)):
            last = i == len(stages) - 1
            enc_stage = SwinStack(c, num_blocks=l, attention=False)
            enc_ch_conv = nn.Identity() if last else nn.Sequential(nn.Conv2d(channels[i], channels[i + 1], 1, 1, 0), nn.AvgPool2d(kernel_size=2))
            dec_stage = SwinStack(c, num_blocks=l)
            dec_ch_conv = nn.Identity() if last else nn.Sequential(nn.Upsample(scale_factor=2), nn.Conv2d(channels[i + 1], channels[i], 1, 1, 0))
            self.encoder_stages.append(UNetBlock(enc_stage, enc_ch_conv))
            self.decoder_stages.insert(0, UNetBlock(dec_stage, dec_ch_conv))
        self.input_channels = input_channels
        self.channels = list(channels)
        self._uniform_time = None          # set by DDPM.sample: every sample shares this timestep
        self._tables = {}
        self._plan = None                  # (key, ctypes plan, keep-alive list) for the native executor
        self._workspace = None
        self.native_forward = True         # inference forwards run through ldm_unet_forward_f32 (one C call per forward)
        self.hoist_films = True            # denoise loops compute the FiLM tables of all their timesteps in the first step
        self.hoist_budget_bytes = 16 << 30 # ... unless those tables would need more workspace than this (2.3 GB at 50 steps, 32x32 latents)
        self._films_key = None             # identity of the tables currently in the workspace (None: not valid)
        self._slot_table = None
        # reduced-precision sampling (ddpm.py:52,75: the reference samples under 16-bit autocast on a GPU).  Opt-in: None (default) keeps
        # every forward exact fp32 whatever `use_autocast` says; torch.bfloat16 (autocast.set_autocast_dtype) makes
        # DDPM.sample(use_autocast=True) run its UNet forwards with bf16 GEMM operands (fp32 accumulate, fp32 residual stream)
        self.autocast_dtype = None
        self._autocast_now = False         # set by DDPM.sample for the duration of an autocast loop
        self._grad_sync = None             # dist.GradSync of the current data-parallel step (dist.train_step), else None
        self._plan16 = None

    def _head_bias(self):
        """decoder_last.bias as the 1x1 head over Cin * s^2 channels needs it: every channel's value once per position of its s x s patch."""
        bias = self.decoder_last.bias.detach()
        return bias if self.stem_size == 1 else bias.repeat_interleave(self.stem_size * self.stem_size)

    def _level_blocks(self, i):
        n = len(self.encoder_stages)
        return list(self.encoder_stages[i].stage.blocks) + list(self.decoder_stages[n - 1 - i].stage.blocks)

    def _pointer_tables(self, i, blocks, dev):
        """Host arrays of the level's proj1/proj2 weight and bias addresses (rebuilt if any moved)."""
        tens = []
        for name in ("proj1", "proj2"):
            tens.append([getattr(b.encodings, name).weight.detach() for b in blocks])
            tens.append([getattr(b.encodings, name).bias.detach() for b in blocks])
        key = (str(dev), tuple(t.data_ptr() for ts in tens for t in ts))
        hit = self._tables.get(i)
        if hit is None or hit[0] != key:
            hit = (key, [ops.pointer_table(ts) for ts in tens])
            self._tables[i] = hit
        return hit[1]

    def _films(self, ctx, h, w, dev):
        """FiLM rows (mul | bias) of EVERY SwinBlock, one pair of grouped GEMM launches per level
        (unet.py:18-21 for all blocks of the level at once; they only depend on t, not on x)."""
        films = {}
        for i, c in enumerate(self.channels):
            blocks = self._level_blocks(i)
            codes = ctx.codes(c, h >> i, w >> i)
            m, g = codes.shape[0], len(blocks)
            w1, b1, w2, b2 = self._pointer_tables(i, blocks, dev)
            hid = torch.empty(g, m, 4 * c, device=dev, dtype=torch.float32)
            ops.gemm(codes, m, 4 * c, 2 * c, None, hid, w_table=w1, bias_table=b1, act=ops.ACT_RELU, groups=g,
                     a_gstride=0, o_gstride=m * 4 * c)
            film = torch.empty(g, m, 2 * c, device=dev, dtype=torch.float32)
            ops.gemm(hid, m, 2 * c, 4 * c, None, film, w_table=w2, bias_table=b2, groups=g,
                     a_gstride=m * 4 * c, o_gstride=m * 2 * c)
            for k, blk in enumerate(blocks):
                films[blk] = film[k]
        return films

    def _native_plan(self, dev):
        """struct ldm_unet_plan over the live parameters (rebuilt when any of them moved / changed version)."""
        import ctypes
        from ._lib import UNetBlockDesc, UNetPlanDesc
        # The plan holds POINTERS to the live parameters, so in-place updates (optimizer steps, load_state_dict)
        # need no rebuild; only the re-laid-out grouped-conv weights are copies (checked by version here), and
        # device / dtype moves drop the plan in _apply().
        # Staleness (three ways a weight can change under the plan): (1) a Parameter OBJECT is replaced (module.weight = ...,
        # load_state_dict(assign=True)): every registration bumps the process-wide _PARAM_GENERATION; (2) a parameter's storage is
        # swapped (p.data = new): the data_ptr of every captured source tensor is part of the key; (3) the packed grouped-conv
        # copies: keyed on the conv weight's version.  In-place writes through ``.data`` (p.data.copy_()) bump no version and are
        # invisible to (3): call invalidate_caches() after such a write.
        if self._plan is not None:
            order, captured = self._plan[3], self._plan[4]
            key = (str(dev), _PARAM_GENERATION[0], tuple(t.data_ptr() for t in captured),
                   tuple(blk.conv.weight._version for blk in order), weights.GENERATION[0],
                   self.decoder_last.bias._version if self.stem_size > 1 else 0)          # (the replicated head bias is a copy)
            if self._plan[0] == key:
                return self._plan[1]
        order = [blk for l in self.encoder_stages for blk in l.stage.blocks] + [blk for l in self.decoder_stages for blk in l.stage.blocks]
        keep, captured = [], []

        def ptr(t):
            src = t
            t = t.detach()
            if not t.is_contiguous():
                t = t.contiguous()
            elif isinstance(src, nn.Parameter):
                captured.append(src)                 # the plan points INTO this parameter's storage
            keep.append(t)
            return t.data_ptr()

        blocks = (UNetBlockDesc * len(order))()
        for bd, blk in zip(blocks, order):
            bd.attention = int(blk.attention_flag)
            bd.shift = blk.self_attention.shift if blk.attention_flag else 0
            bd.conv_w, bd.conv_b = ptr(blk._conv_weight()), ptr(blk.conv.bias)
            enc = blk.encodings
            bd.enc_w1, bd.enc_b1, bd.enc_w2, bd.enc_b2 = ptr(enc.proj1.weight), ptr(enc.proj1.bias), ptr(enc.proj2.weight), ptr(enc.proj2.bias)
            for k, r in enumerate([blk.ffn.general] + list(blk.ffn.experts)):
                bd.a_w[k], bd.a_b[k] = ptr(r.a.weight), ptr(r.a.bias)
                bd.b_w[k], bd.b_b[k] = ptr(r.b.weight), ptr(r.b.bias)
                bd.c_w[k], bd.c_b[k] = ptr(r.c.weight), ptr(r.c.bias)
            if blk.attention_flag:
                att = blk.self_attention.attention
                bd.in_w, bd.in_b = ptr(att.in_proj_weight), ptr(att.in_proj_bias)
                bd.out_w, bd.out_b = ptr(att.out_proj.weight), ptr(att.out_proj.bias)
        plan = UNetPlanDesc()
        n = len(self.encoder_stages)
        plan.levels, plan.input_channels, plan.window, plan.nblocks = n, self.input_channels * self.stem_size ** 2, 6, len(order)
        plan.eps = order[0].norm.eps
        for i in range(n):
            plan.channels[i] = self.channels[i]
            plan.enc_blocks[i] = len(self.encoder_stages[i].stage.blocks)
            plan.dec_blocks[i] = len(self.decoder_stages[n - 1 - i].stage.blocks)
            pf, tf = sinusoidal.FREQS.get(self.channels[i], dev)
            plan.pos_freq[i], plan.time_freq[i] = ptr(pf), ptr(tf)
            if i < n - 1:
                down = self.encoder_stages[i].ch_conv[0]
                up = self.decoder_stages[n - 1 - i].ch_conv[1]
                plan.down_w[i], plan.down_b[i] = ptr(down.weight), ptr(down.bias)
                plan.up_w[i], plan.up_b[i] = ptr(up.weight), ptr(up.bias)
        plan.stem_w, plan.stem_b = ptr(self.encoder_first.weight), ptr(self.encoder_first.bias)
        plan.head_w, plan.head_b = ptr(self.decoder_last.weight), ptr(self.decoder_last.bias if self.stem_size == 1 else self._head_bias())
        plan.blocks = ctypes.cast(blocks, ctypes.POINTER(UNetBlockDesc))
        keep.append(blocks)
        key = (str(dev), _PARAM_GENERATION[0], tuple(t.data_ptr() for t in captured), tuple(blk.conv.weight._version for blk in order),
               weights.GENERATION[0], self.decoder_last.bias._version if self.stem_size > 1 else 0)
        self._plan = (key, plan, keep, order, captured)
        return plan

    def _gemm_weights16(self, order):
        """(parameter, [rows, cols] view) of every weight the bf16 forward reads, block by block."""
        out = []
        for blk in order:
            out.append(("conv", blk, blk._conv_weight()))
            for k, r in enumerate([blk.ffn.general] + list(blk.ffn.experts)):
                out += [(("a", k), blk, w2d(r.a)), (("b", k), blk, w2d(r.b)), (("c", k), blk, w2d(r.c))]
            if blk.attention_flag:
                att = blk.self_attention.attention
                out += [("in", blk, att.in_proj_weight.detach()), ("out", blk, att.out_proj.weight.detach())]
        return out

    def _native_plan16(self, dev, check=True):
        """struct ldm_unet_plan_bf16: bf16 copies of the GEMM weights (one flat buffer), re-cast when any weight changed.
        ``check=False`` (later steps of one denoise loop: weights cannot have moved since its first step) skips the version scan."""
        import ctypes
        from ._lib import UNetBlock16Desc, UNetPlan16Desc
        if not check and self._plan16 is not None:
            return self._plan16[1]
        self._native_plan(dev)
        order = self._plan[3]
        srcs = [p for l in (list(self.encoder_stages) + list(self.decoder_stages)) for blk in l.stage.blocks for p in blk.parameters()]
        key = (self._plan[0], sum(p._version for p in srcs), tuple(blk.conv.weight.data_ptr() for blk in order))
        if self._plan16 is not None and self._plan16[0] == key:
            return self._plan16[1]
        for c in self.channels:
            if c % 64:
                raise NotImplementedError("bf16 sampling needs every stage width to be a multiple of 64 (got %r)" % (self.channels,))
        items = self._gemm_weights16(order)
        total = sum(w.numel() for _, _, w in items)
        buf = torch.empty(total, device=dev, dtype=torch.bfloat16)
        blocks = (UNetBlock16Desc * len(order))()
        index = {blk: i for i, blk in enumerate(order)}
        off = 0
        for name, blk, w in items:
            n = w.numel()
            dst = buf[off:off + n]
            ops.cast_bf16(w.contiguous().reshape(-1), dst)
            off += n
            bd = blocks[index[blk]]
            if name == "conv":
                bd.conv_w = dst.data_ptr()
            elif name == "in":
                bd.in_w = dst.data_ptr()
            elif name == "out":
                bd.out_w = dst.data_ptr()
            else:
                getattr(bd, name[0] + "_w")[name[1]] = dst.data_ptr()
        plan16 = UNetPlan16Desc()
        plan16.nblocks = len(order)
        plan16.blocks = ctypes.cast(blocks, ctypes.POINTER(UNetBlock16Desc))
        self._plan16 = (key, plan16, (buf, blocks))
        return plan16

    def invalidate_caches(self):
        """Drop every derived copy of the weights (native plan, pointer tables, packed grouped-conv filters).  Needed only after
        writing weights in place through ``.data`` (p.data.copy_(), EMA swaps, clipping), which bumps no version counter; updates
        through the Parameter itself (optimizer steps, load_state_dict, p.copy_ under no_grad) are tracked automatically."""
        weights.bump()                     # every version-keyed copy (transposed / bf16 weights of the training step too)
        self._plan = None
        self._plan16 = None
        self._tables = {}
        for stack in list(self.encoder_stages) + list(self.decoder_stages):
            for blk in stack.stage.blocks:
                blk._packed = None

    def _apply(self, fn, *args, **kwargs):
        self._plan = None                  # parameters are about to be replaced (.to / .cuda / .float)
        self._plan16 = None
        self._tables = {}
        return super()._apply(fn, *args, **kwargs)

    def _forward_native(self, x, ctx):
        """The whole forward as one C call (csrc/unet_exec.cpp): same launches, same order, same bits."""
        import ctypes
        from . import _lib
        b, cin, h, w = x.shape
        dev = x.device
        plan = self._native_plan(dev)
        order = self._plan[3]
        dec = (ctypes.c_int * len(order))()
        for k, blk in enumerate(order):                          # Python-RNG draws, reference order
            d = blk.draw()
            dec[k] = -1 if d is None else d[0] * 4 + d[1]
        lib = _lib.load()
        # denoise loop (DDPM.sample hands over its whole timestep list): the FiLM tables depend on t, never on x, so the first step
        # computes them for ALL timesteps in one pair of GEMM launches per level (the weights -- 0.74 GB -- are read once per loop
        # instead of once per step) and every step selects its rows through slot[b] = step index
        sched = self._uniform_time if (self.hoist_films and isinstance(self._uniform_time, tuple) and len(self._uniform_time) == 4) else None
        if sched is not None:                                    # tables of the whole loop: (codes 2C + hidden 4C x blocks + rows 2C x blocks) floats
            n_t = sched[2].numel()
            need_f = sum(n_t * (h >> i) * (w >> i) * c * (2 + 6 * len(self._level_blocks(i))) for i, c in enumerate(self.channels))
            if 4 * need_f > self.hoist_budget_bytes:             # very long schedules / large latents: stay with the per-step form
                sched = None
        if sched is not None:
            t_unique, step = sched[2], int(sched[3])
            nt = t_unique.numel()
            if self._slot_table is None or self._slot_table.shape != (nt, b) or self._slot_table.device != dev:
                self._slot_table = torch.arange(nt, dtype=torch.int32, device=dev).repeat_interleave(b).reshape(nt, b).contiguous()
            slot = self._slot_table[step]
        else:
            t_unique, nt, slot = ctx.t_unique, ctx.t_unique.numel(), ctx.slot
            self._films_key = None
        need = lib.ldm_unet_workspace_bytes(ctypes.byref(plan), b, h, w, nt)
        if need == 0:
            raise ValueError("UNet: input %dx%d is not divisible by 2**%d" % (h, w, len(self.channels) - 1))
        if self._workspace is None or self._workspace.numel() < need or self._workspace.device != dev:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=dev)
            self._films_key = None
        key = None if sched is None else (t_unique.data_ptr(), nt, b, h, w, self._workspace.data_ptr(), id(plan), self._plan[0])
        ready = int(key is not None and key == self._films_key)
        out = torch.empty(b, cin, h, w, device=dev, dtype=torch.float32)
        x = x.contiguous().float()
        if self._autocast_now and self.autocast_dtype is torch.bfloat16:
            plan16 = self._native_plan16(dev, check=not ready)
            _lib.check(lib.ldm_unet_forward_bf16(ctypes.byref(plan), ctypes.byref(plan16), x.data_ptr(), t_unique.data_ptr(), nt,
                                                 None if slot is None else slot.data_ptr(), dec, b, h, w,
                                                 self._workspace.data_ptr(), need, out.data_ptr(), ready,
                                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "ldm_unet_forward_bf16")
        else:
            _lib.check(lib.ldm_unet_forward_ex_f32(ctypes.byref(plan), x.data_ptr(), t_unique.data_ptr(), nt,
                                                   None if slot is None else slot.data_ptr(), dec, b, h, w,
                                                   self._workspace.data_ptr(), need, out.data_ptr(), ready,
                                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "ldm_unet_forward_ex_f32")
        self._films_key = key
        return out

    def forward(self, x, time, condition=None):
        s = self.stem_size
        if s == 1:
            return self._forward_1x1(x, time)
        if x.shape[-1] % s or x.shape[-2] % s:
            raise ValueError("UNet: input %dx%d is not divisible by stem_size %d" % (x.shape[-2], x.shape[-1], s))
        # channel ci * s^2 + dy * s + dx of the re-tiled image is pixel (dy, dx) of the patch: the order of encoder_first.weight[:, ci, dy, dx]
        return F.pixel_shuffle(self._forward_1x1(F.pixel_unshuffle(x, s), time), s)

    def _forward_1x1(self, x, time):
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            # training step: same kernels + saved activations, hand-written backward (train.py).  Like the reference's autograd this
            # keeps every block's activations alive (~21 C floats per pixel and block: 63 GB fp32 / 43 GB bf16 at the cfg-5 shape)
            # whether or not the module is in train() mode -- wrap pure inference in torch.no_grad(), as ddpm.sample does.
            from .train import UNetFunction
            params = [p for p in self.parameters() if p.requires_grad]
            return UNetFunction.apply(self, x, time, *params)
        b, cin, h, w = x.shape
        dev = x.device
        if not x.is_cuda:
            from ._lib import LdmHipUnavailable
            raise LdmHipUnavailable("x must be a GPU tensor: the HIP path is the only implementation (no CPU fallback)")
        ctx = TimeContext(time, b, dev, uniform=self._uniform_time)
        if self.native_forward or self._autocast_now:          # the bf16 mode exists in the native executor only
            return self._forward_native(x, ctx)
        # Python-RNG decisions of all 36 blocks in execution order (identical draw order to the
        # reference, which draws them lazily inside each block: nothing else touches `random`).
        order = [blk for l in self.encoder_stages for blk in l.stage.blocks] + \
                [blk for l in self.decoder_stages for blk in l.stage.blocks]
        decisions = {blk: blk.draw() for blk in order}
        films = self._films(ctx, h, w, dev)
        plan = {blk: (decisions[blk], films[blk]) for blk in order}
        c0 = self.channels[0]
        rows = torch.empty(b * h * w, c0, device=dev, dtype=torch.float32)
        ops.stem_nchw(x.contiguous().float(), w2d(self.encoder_first), self.encoder_first.bias.detach(), rows, b, cin, h * w, c0)
        skips = []
        n = len(self.encoder_stages)
        for i, l in enumerate(self.encoder_stages):
            rows = l.stage.forward_rows(rows, (b, h, w), ctx, plan)
            if i == n - 1:
                skips.insert(0, None)                                  # unet.py:94-95 (adds the integer 0)
            else:
                skips.insert(0, rows)
                conv = l.ch_conv[0]
                cn = conv.weight.shape[0]
                # AvgPool2d(Conv1x1(x)) == Conv1x1(AvgPool2d(x)) (both linear); pooled first: 4x fewer FLOPs
                pooled = torch.empty(b * (h // 2) * (w // 2), rows.shape[1], device=dev, dtype=torch.float32)
                ops.avgpool2(rows, pooled, b, h, w, rows.shape[1])
                h, w = h // 2, w // 2
                rows = torch.empty(b * h * w, cn, device=dev, dtype=torch.float32)
                ops.gemm(pooled, b * h * w, cn, pooled.shape[1], [w2d(conv)], rows, biases=[conv.bias.detach()])
        for i, (l, s) in enumerate(zip(self.decoder_stages, skips)):
            if not isinstance(l.ch_conv, nn.Identity):
                conv = l.ch_conv[1]
                cn = conv.weight.shape[0]
                # Conv1x1(nearest_up2(x)) == nearest_up2(Conv1x1(x)) exactly; the skip is added in the epilogue
                up = torch.empty(b * 4 * h * w, cn, device=dev, dtype=torch.float32)
                ops.gemm(rows, b * h * w, cn, rows.shape[1], [w2d(conv)], up, biases=[conv.bias.detach()],
                         addend=s, o_mode=ops.O_UP2, out_hw=(h, w))
                h, w = 2 * h, 2 * w
                rows = up
            elif s is not None:
                raise RuntimeError("unexpected skip at an Identity ch_conv")
            rows = l.stage.forward_rows(rows, (b, h, w), ctx, plan)
        out = torch.empty(b, cin, h, w, device=dev, dtype=torch.float32)
        wl = self.decoder_last.weight.detach().reshape(c0, cin)         # ConvTranspose2d weight [C0, Cin, 1, 1]
        ops.head_nchw(rows, wl, self._head_bias(), out, b, c0, h * w, cin)
        return out
