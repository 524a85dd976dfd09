"""Sin/cos position and time codes (reference: sinusoidal.py:6-41).

Same class names and constructor arguments as the reference.  The frequency
tables are computed on the host with the reference's own expressions (so the
large-argument trigonometry sees bit-identical frequencies); the codes themselves
are produced on the GPU by ``ldm_sincos_embed_f32``.
"""
import torch
import torch.nn as nn

from . import ops


def position_frequencies(channels):
    """sinusoidal.py:15: 1 / 2**(k/(C/4))."""
    q = channels // 4
    return 1 / (2 ** (torch.arange(q) / q))


def time_frequencies(channels, max_timesteps=10000):
    """sinusoidal.py:34: 1 / max_timesteps**(k/(C/2))."""
    half = channels // 2
    return 1 / (max_timesteps ** (torch.arange(half) / half))


class _FreqCache:
    """Device copies of the two frequency tables for one channel count."""

    def __init__(self):
        self._tabs = {}

    def get(self, channels, device, max_timesteps=10000):
        key = (channels, str(device), max_timesteps)
        if key not in self._tabs:
            self._tabs[key] = (position_frequencies(channels).float().to(device),
                               time_frequencies(channels, max_timesteps).float().to(device))
        return self._tabs[key]


FREQS = _FreqCache()


def embed(t_unique, height, width, channels, max_timesteps=10000):
    """cat[PositionalEncoding2d, TimeEncoding2d] as rows: [nT*H*W, 2C] (unet.py:19)."""
    pf, tf = FREQS.get(channels, t_unique.device, max_timesteps)
    out = torch.empty(t_unique.numel() * height * width, 2 * channels, device=t_unique.device, dtype=torch.float32)
    return ops.sincos_embed(t_unique, height, width, channels, pf, tf, out)


class PositionalEncoding2d(nn.Module):
    def __init__(self, channels, return_encoding_only=False):
        super().__init__()
        self.channels = channels
        self.return_encoding_only = return_encoding_only

    def forward(self, x):
        n, c, h, w = x.shape
        t = torch.zeros(1, dtype=torch.int64, device=x.device)
        rows = embed(t, h, w, c)[:, :c]                          # [H*W, C]
        emb = rows.reshape(1, h, w, c).permute(0, 3, 1, 2).expand(n, c, h, w)
        return emb if self.return_encoding_only else x + emb


class TimeEncoding2d(nn.Module):
    def __init__(self, channels, max_timesteps=10000, return_encoding_only=False):
        super().__init__()
        self.channels = channels
        self.max_timesteps = max_timesteps
        self.return_encoding_only = return_encoding_only

    def forward(self, x, t):
        n, c, h, w = x.shape
        rows = embed(t.to(torch.int64).contiguous(), 1, 1, c, self.max_timesteps)[:, c:]   # [B, C]
        emb = rows.reshape(n, c, 1, 1).expand(n, c, h, w)
        return emb if self.return_encoding_only else x + emb
