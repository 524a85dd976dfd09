"""Block primitives (reference: modules.py:7-36), GPU-only.

Class names, constructor arguments, parameter names and registration order are
the reference's, so ``state_dict`` keys, default initialisation and checkpoint
loading are drop-in.  ``nn.Conv2d`` objects are kept purely as parameter holders
(their ``forward`` is never called): all arithmetic goes through the HIP library.

Internally activations are channels-last rows ``[M = B*H*W, C]``; the public
``forward`` methods accept/return the reference's NCHW tensors.
"""
import random

import torch
import torch.nn as nn

from . import ops


def to_rows(x):
    """NCHW tensor -> ([M, C] channels-last rows, (B, H, W))."""
    b, c, h, w = x.shape
    out = torch.empty(b * h * w, c, device=x.device, dtype=torch.float32)
    ops.nchw_to_nhwc(x.contiguous().float(), out, b, c, h * w)
    return out, (b, h, w)


def from_rows(rows, shape):
    b, h, w = shape
    c = rows.shape[1]
    out = torch.empty(b, c, h, w, device=rows.device, dtype=torch.float32)
    ops.nhwc_to_nchw(rows, out, b, c, h * w)
    return out


def w2d(conv):
    """[Cout, Cin, 1, 1] conv weight viewed as the [N, K] GEMM operand (no copy)."""
    w = conv.weight
    return w.detach().reshape(w.shape[0], -1)


class ReGLU(nn.Module):
    """c(a(x) * relu(b(x)))  (modules.py:7-15)."""

    def __init__(self, channels, ffn_mul=4):
        super().__init__()
        self.a = nn.Conv2d(channels, channels * ffn_mul, 1, 1, 0)
        self.b = nn.Conv2d(channels, channels * ffn_mul, 1, 1, 0)
        self.act = nn.ReLU()
        self.c = nn.Conv2d(channels * ffn_mul, channels, 1, 1, 0)

    def forward(self, x):
        rows, shape = to_rows(x)
        return from_rows(reglu_sum(rows, [self]), shape)


def reglu_sum(rows, reglus, addend=None, out=None):
    """sum_i c_i(a_i(x) * relu(b_i(x))) (+ addend) for up to 4 ReGLUs of equal width.

    Two launches: a gated GEMM producing the stacked hidden [M, n*F], then one
    GEMM over the stacked K with per-expert weight pointers.  Expert
    weights are addressed in place -- selecting experts costs nothing.
    """
    m, c = rows.shape
    n = len(reglus)
    f = reglus[0].a.weight.shape[0]
    hidden = torch.empty(m, n * f, device=rows.device, dtype=torch.float32)
    ops.gemm(rows, m, n * f, c, [w2d(r.a) for r in reglus], hidden,
             weights2=[w2d(r.b) for r in reglus], biases=[r.a.bias.detach() for r in reglus],
             biases2=[r.b.bias.detach() for r in reglus], seg_mode=ops.SEG_N, act=ops.ACT_GATE)
    if out is None:
        out = torch.empty(m, c, device=rows.device, dtype=torch.float32)
    ops.gemm(hidden, m, c, n * f, [w2d(r.c) for r in reglus], out, biases=[r.c.bias.detach() for r in reglus],
             seg_mode=ops.SEG_K, addend=addend)
    return out


class ChannelNorm(nn.Module):
    """Per-pixel normalisation over C, unbiased variance, eps in the sqrt (modules.py:18-25)."""

    def __init__(self, channels, eps=1e-4):
        super().__init__()
        self.eps = eps

    def forward(self, x):
        rows, (b, h, w) = to_rows(x)
        c = rows.shape[1]
        film = torch.zeros(h * w, 2 * c, device=x.device, dtype=torch.float32)
        film[:, :c] = 1.0                                   # identity FiLM
        out = torch.empty_like(rows)
        ops.channelnorm_film(rows, film, None, out, b, h * w, c, self.eps)
        return from_rows(out, (b, h, w))


class RandomMoE(nn.Module):
    """general + 2 of 4 experts drawn with Python's global ``random`` (modules.py:28-36)."""

    def __init__(self, channels, ffn_mul=1, num_experts=4):
        super().__init__()
        self.general = ReGLU(channels, ffn_mul=ffn_mul)
        self.experts = nn.ModuleList([ReGLU(channels, ffn_mul=ffn_mul) for _ in range(num_experts)])

    def pick(self):
        # identical RNG consumption to random.sample(list(self.experts), 2): CPython's
        # sample() only looks at len(population) and k.
        return random.sample(range(len(self.experts)), 2)

    def forward_rows(self, rows, addend=None, out=None, picks=None):
        if picks is None:
            picks = self.pick()
        return reglu_sum(rows, [self.general] + [self.experts[i] for i in picks], addend=addend, out=out)

    def forward(self, x):
        rows, shape = to_rows(x)
        return from_rows(self.forward_rows(rows), shape)
