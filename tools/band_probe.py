"""Tile-order probe: the GEMM family's banded order (gemm_common.h, kBandM M-tiles per band) against other band heights, on the level-0 / 1 / 2
gated GEMMs of the sampling pass (bf16 and fp32) -- the launches whose HBM fetch is ~3x algorithmic.
    for b in 1 2 4 16; do hipcc ... -DLDM_BAND_M=$b -o ldm_image_generator_amd/variants/libldm_band$b.so ...; done
    LDM_HIP_LIB=.../libldm_band4.so python tools/band_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_image_generator_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16


def timed(fn, it=10):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / it * 1e3)
    return best


print("lib:", os.environ.get("LDM_HIP_LIB", "default"))
for s, (C, R) in enumerate([(128, 32), (256, 16), (512, 8)]):
    M = 256 * R * R
    x = torch.randn(M, C, device=dev)
    x16 = x.to(BF)
    wa = [torch.randn(C, C, device=dev) * C ** -0.5 for _ in range(3)]
    wb = [torch.randn(C, C, device=dev) * C ** -0.5 for _ in range(3)]
    bs = [torch.randn(C, device=dev) for _ in range(3)]
    hid = torch.empty(M, 3 * C, device=dev)
    hid16 = torch.empty(M, 3 * C, device=dev, dtype=BF)
    wa16, wb16 = [w.to(BF) for w in wa], [w.to(BF) for w in wb]
    t32 = timed(lambda: ops.gemm(x, M, 3 * C, C, wa, hid, weights2=wb, biases=bs, biases2=bs, act=ops.ACT_GATE))
    t16 = timed(lambda: ops.gemm_bf16_gate_fwd(x16, M, 3 * C, C, wa16, wb16, hid16, biases_a=bs, biases_b=bs))
    wc16 = [torch.randn(C, C, device=dev).to(BF) * C ** -0.5 for _ in range(3)]
    y = torch.randn(M, C, device=dev)
    tc = timed(lambda: ops.gemm_bf16(hid16, M, C, 3 * C, wc16, y, biases=bs, seg_mode=ops.SEG_K, addend=y))
    wq16 = torch.randn(3 * C, C, device=dev).to(BF) * C ** -0.5
    q16 = torch.empty(M, 3 * C, device=dev, dtype=BF)
    tq = timed(lambda: ops.gemm_bf16(x16, M, 3 * C, C, [wq16], q16))
    print("level %d (M=%d C=%d): fp32 gate %7.1f us | bf16 gate %7.1f us | bf16 K-seg c-GEMM %7.1f us | bf16 QKV %7.1f us" % (s, M, C, t32, t16, tc, tq), flush=True)
