"""Per-shape rates of the bf16 training GEMMs at the BASELINE cfg-5 per-GPU shapes (B = 128, latents 64x64: M = 524288 >> level).
    python tools/bf16_bench.py [--batch 128] [--latent 64] [--iters 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_image_generator_amd import ops, train  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--latent", type=int, default=64)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--levels", default="0,1,2,3")
args = ap.parse_args()
dev = torch.device("cuda:0")
BF = torch.bfloat16


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters * 1e-3


print("%-34s %9s %9s %9s" % ("shape", "us", "TFLOP/s", "GB/s(algo)"))
for lvl, c in enumerate([128, 256, 512, 1024]):
    if str(lvl) not in args.levels.split(","):
        continue
    m = args.batch * (args.latent >> lvl) ** 2
    g = torch.Generator().manual_seed(c)

    def rnd(*s, dt=BF):
        return torch.randn(*s, generator=g).to(dt).to(dev)

    cases = [("NT a_pre  N=3C K=C  ->bf16", 3 * c, c, BF, False), ("NT gemm2  N=C K=3C  +=f32", c, 3 * c, torch.float32, True),
             ("NT film   N=2C K=4C ->f32", 2 * c, 4 * c, torch.float32, False), ("NT dh     N=4C K=2C ->bf16", 4 * c, 2 * c, BF, False)]
    for name, n, k, odt, add in cases:
        a, w = rnd(m, k), rnd(n, k)
        out = torch.zeros(m, n, device=dev, dtype=odt)
        dt = timed(lambda: ops.gemm_bf16(a, m, n, k, [w], out, addend=out if add else None))
        by = 2 * (m * k + n * k) + m * n * (2 if odt == BF else 4) * (2 if add else 1)
        print("C=%-4d M=%-7d %-22s %9.1f %9.1f %9.0f" % (c, m, name, dt * 1e6, 2.0 * m * n * k / dt / 1e12, by / dt / 1e9))
        del a, w, out
    for name, n, k in [("TN dWa    N=3C K=C", 3 * c, c), ("TN dWc    N=C K=3C", c, 3 * c), ("TN dW2    N=2C K=4C", 2 * c, 4 * c)]:
        dy, x = rnd(m, n), rnd(m, k)
        s = train.tn16_splits(n, k, m)
        parts = torch.empty(s, n, k, device=dev)
        cs = torch.empty(s, n, device=dev)
        dt = timed(lambda: ops.gemm_tn_bf16(dy, x, parts, m, n, k, s, colsum=cs))
        print("C=%-4d M=%-7d %-22s %9.1f %9.1f %9.0f  (splits %d)" % (c, m, name, dt * 1e6, 2.0 * m * n * k / dt / 1e12, (2 * m * (n + k) + 4 * n * k * s) / dt / 1e9, s))
        del dy, x, parts
