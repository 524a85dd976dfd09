"""Window attention (reference: attention.py:5-98), GPU-only.

``nn.MultiheadAttention`` is kept only as the parameter holder (keys
``attention.in_proj_weight`` ... as in the reference's checkpoints).  The
computation is: one GEMM for the packed in-projection of the UNPADDED tokens, the
fused window kernel (pad / roll / split / mask / softmax / AV / un-roll / crop as
index arithmetic), one GEMM for the out-projection.
"""
import torch
import torch.nn as nn

from . import ops
from .modules import from_rows, to_rows


class WindowAttention(nn.Module):
    def __init__(self, channels=512, n_heads=8, window_size=4, shift=0):
        super().__init__()
        if channels % n_heads != 0 or channels // n_heads != 32:
            raise NotImplementedError("the HIP attention core is specialised for head_dim == 32 (unet.py:26)")
        self.attention = nn.MultiheadAttention(channels, n_heads, batch_first=True)
        self.window_size = window_size
        self.shift = shift

    def forward_rows(self, rows, shape, addend=None, out=None):
        """rows [M, C] -> out-projection of the attention (+ addend)."""
        b, h, w = shape
        m, c = rows.shape
        att = self.attention
        qkv = torch.empty(m, 3 * c, device=rows.device, dtype=torch.float32)
        ops.gemm(rows, m, 3 * c, c, [att.in_proj_weight.detach()], qkv, biases=[att.in_proj_bias.detach()])
        ctx = torch.empty(m, c, device=rows.device, dtype=torch.float32)
        ops.window_attention(qkv, att.in_proj_bias.detach(), rows, ctx, b, h, w, c, self.window_size, self.shift)
        if out is None:
            out = torch.empty(m, c, device=rows.device, dtype=torch.float32)
        ops.gemm(ctx, m, c, c, [att.out_proj.weight.detach()], out, biases=[att.out_proj.bias.detach()], addend=addend)
        return out

    def forward(self, x):
        rows, shape = to_rows(x)
        return from_rows(self.forward_rows(rows, shape), shape)


class CrossAttention(nn.Module):
    """Dead code in the reference (attention.py:87-98: never called by UNet.forward,
    and its forward has no return).  Kept so the 4 parameter tensors per attention
    block exist in ``state_dict`` exactly as in the reference's checkpoints."""

    def __init__(self, channels=512, n_heads=8):
        super().__init__()
        self.attention = nn.MultiheadAttention(channels, n_heads, batch_first=True)

    def forward(self, x, c):
        return None         # attention.py:92-98 falls off the end without a return
