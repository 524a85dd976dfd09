"""bf16 NT GEMM: 128 x 128 stream kernel vs 256 x 256 ring kernel (gemm_bf16_ring.hip) on the training step's shapes.
    python tools/ring_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_image_generator_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16


def timed(fn, it=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3


# (M, N, K, output dtype, K-segments + in-place addend, segments)
shapes = [(32768, 2048, 1024, BF, 0, 1), (32768, 1024, 2048, torch.float32, 0, 1), (32768, 512, 1536, torch.float32, 1, 3), (32768, 1536, 512, torch.float32, 0, 1),
          (131072, 1024, 512, BF, 0, 1), (131072, 512, 1024, torch.float32, 0, 1), (131072, 256, 768, torch.float32, 1, 3), (131072, 768, 256, BF, 0, 3),
          (524288, 512, 256, BF, 0, 1), (524288, 256, 512, torch.float32, 0, 1), (8192, 4096, 2048, BF, 0, 1), (8192, 2048, 4096, torch.float32, 0, 1),
          (8192, 1024, 3072, torch.float32, 1, 3)]
for (m, n, k, odt, add, nseg) in shapes:
    a = torch.randn(m, k, device=dev).to(BF)
    if add:
        ws = [torch.randn(n, k // nseg, device=dev).to(BF) for _ in range(nseg)]
        sm = ops.SEG_K
    else:
        ws = [torch.randn(n // nseg, k, device=dev).to(BF) for _ in range(nseg)]
        sm = ops.SEG_N
    out = torch.zeros(m, n, device=dev, dtype=odt)
    res = []
    for mode in (0, 2):
        ops.gemm_ring(mode)
        res.append(timed(lambda: ops.gemm_bf16(a, m, n, k, ws, out, seg_mode=sm, addend=out if add else None)))
    ops.gemm_ring(1)
    fl = 2.0 * m * n * k
    print("M=%-7d N=%-5d K=%-5d %-5s %s  stream %7.1f us %6.0f TF | ring %7.1f us %6.0f TF  (x%.2f)"
          % (m, n, k, "bf16" if odt == BF else "f32", "add" if add else "   ", res[0] * 1e6, fl / res[0] / 1e12, res[1] * 1e6, fl / res[1] / 1e12, res[0] / res[1]))
