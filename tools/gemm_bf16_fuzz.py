"""Randomised parity sweep of ldm_gemm_bf16: stream kernel vs ring kernel (bit-identical) vs fp64 on the same bf16 operands.
    python tools/gemm_bf16_fuzz.py [cases] [seed]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_image_generator_amd import ops as o  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
g = torch.Generator().manual_seed(rng.randrange(1 << 30))


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).float()


bad = 0
for ci in range(cases):
    M = rng.choice([256, 512, 768, 1024, 2048, 4096, 300, 1000])            # the last two never take the ring kernel
    nseg = rng.choice([1, 1, 2, 3])
    seg_mode = rng.choice([o.SEG_N, o.SEG_K])
    unit = rng.choice([64, 128, 256, 512])
    if seg_mode == o.SEG_N:
        N, K = unit * nseg, rng.choice([64, 128, 192, 256, 512, 1024])
    else:
        N, K = rng.choice([64, 128, 256, 512, 768]), unit * nseg
    out16 = rng.random() < 0.5
    use_add = (not out16) and rng.random() < 0.5
    act = rng.choice([o.ACT_NONE, o.ACT_RELU, o.ACT_LRELU])
    a = rnd(M, K).to(BF).to(dev)
    if seg_mode == o.SEG_N:
        ws = [rnd(N // nseg, K, scale=K ** -0.5).to(BF).to(dev) for _ in range(nseg)]
        bs = [rnd(N // nseg).to(dev) if rng.random() < 0.8 else None for _ in range(nseg)]
        ref = torch.cat([a.double() @ w.double().t() + (0 if b is None else b.double()) for w, b in zip(ws, bs)], 1)
    else:
        c = K // nseg
        ws = [rnd(N, c, scale=K ** -0.5).to(BF).to(dev) for _ in range(nseg)]
        bs = [rnd(N).to(dev) if rng.random() < 0.8 else None for _ in range(nseg)]
        ref = sum(a[:, i * c:(i + 1) * c].double() @ ws[i].double().t() + (0 if bs[i] is None else bs[i].double()) for i in range(nseg))
    if act == o.ACT_RELU:
        ref = torch.relu(ref)
    elif act == o.ACT_LRELU:
        ref = torch.where(ref > 0, ref, ref * 0.1)
    base = rnd(M, N).to(dev) if use_add else None
    if use_add:
        ref = ref + base.double()
    outs = []
    old = o.gemm_ring(0)
    for mode in (0, 3):
        o.gemm_ring(mode)
        out = base.clone() if use_add else torch.full((M, N), float("nan"), device=dev, dtype=BF if out16 else torch.float32)
        o.gemm_bf16(a, M, N, K, ws, out, biases=bs, seg_mode=seg_mode, act=act, slope=0.1, addend=out if use_add else None)
        outs.append(out)
    o.gemm_ring(old)
    err = float((outs[1].double() - ref).norm() / ref.norm().clamp_min(1e-30))
    ok = torch.equal(outs[0], outs[1]) and err < (4e-3 if out16 else 1e-5)
    if not ok:
        bad += 1
        print("MISMATCH case %d: M=%d N=%d K=%d nseg=%d seg_mode=%d act=%d add=%s out16=%s: err %.2e, stream==ring %s"
              % (ci, M, N, K, nseg, seg_mode, act, use_add, out16, err, torch.equal(outs[0], outs[1])), flush=True)
print("%d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
