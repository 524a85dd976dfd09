"""Window-attention core, forward and backward, at the UNet's four decoder levels.
    python tools/attn_bench.py [--batch 128] [--latent 64]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_image_generator_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--latent", type=int, default=64)
args = ap.parse_args()
dev = torch.device("cuda:0")


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for s in range(4):
    C, R, B = 128 << s, args.latent >> s, args.batch
    M = B * R * R
    qkv = torch.randn(M, 3 * C, device=dev)
    bias = torch.randn(3 * C, device=dev)
    xf = torch.randn(M, C, device=dev)
    dctx = torch.randn(M, C, device=dev)
    out = torch.empty(M, C, device=dev)
    dqkv = torch.empty(M, 3 * C, device=dev)
    dbp = torch.empty(3 * C, device=dev)
    for shift in (0, 3):
        tf = timeit(lambda: ops.window_attention(qkv, bias, xf, out, B, R, R, C, 6, shift))
        tb = timeit(lambda: ops.window_attention_bwd(qkv, bias, xf, dctx, dqkv, dbp, B, R, R, C, 6, shift))
        old = ops.window_attention_bwd_mfma(0)
        ts = timeit(lambda: ops.window_attention_bwd(qkv, bias, xf, dctx, dqkv, dbp, B, R, R, C, 6, shift))
        ops.window_attention_bwd_mfma(old)
        print("C %4d R %2d shift %d: fwd %.3f ms (%.0f GB/s), bwd %.3f ms (%.0f GB/s; scalar kernel %.3f ms)"
              % (C, R, shift, tf, M * C * 16 / tf / 1e6, tb, M * C * 32 / tb / 1e6, ts), flush=True)
