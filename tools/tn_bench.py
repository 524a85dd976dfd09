"""Weight-gradient (TN) GEMM at the training shapes of BASELINE cfg 5 (B = 128, latents 64x64).
    python tools/tn_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_image_generator_amd import ops  # noqa: E402
from ldm_image_generator_amd.train import _tn_splits  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=6):
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
    C, R = 128 << s, 64 >> s
    M = 128 * R * R
    for (N, K, what) in ((C, 3 * C, "dWc"), (3 * C, C, "dWa")):
        a = torch.randn(M, N, device=dev)
        b = torch.randn(M, K, device=dev)
        S = _tn_splits(N, K, M)
        parts = torch.empty(S, N, K, device=dev)
        cs = torch.empty(S, N, device=dev)
        ms = timeit(lambda: ops.gemm_tn(a, b, parts, M, N, K, S, colsum=cs))
        print("s%d %s M=%d N=%d K=%d splits=%d: %.3f ms  %.1f TFLOP/s  (%.2f TB/s operand reads if uncached)"
              % (s, what, M, N, K, S, ms, 2.0 * M * N * K / ms / 1e9, (M * N * (K // 128) + M * K * (N // 128)) * 4 / ms / 1e9), flush=True)
