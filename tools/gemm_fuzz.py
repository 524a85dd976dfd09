"""Randomised parity sweep of ldm_gemm_f32 (all schedules) against fp64: python tools/gemm_fuzz.py [cases] [seed]."""
import os
import random
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_image_generator_amd import ops as o  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
g = torch.Generator().manual_seed(rng.randrange(1 << 30))


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).float()


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def run_all(fn):
    """schedules 0 (tile), 1 (stream; ring kernel off), 2 (split consumer), then 1 with the ring kernel forced on at most eight
    workgroups (many tiles per workgroup) for every shape it takes"""
    outs = []
    ring = o.gemm_ring(0)
    for v in (0, 1, 2):
        old = o.gemm_variant(v)
        outs.append(fn())
        o.gemm_variant(old)
    o.gemm_ring(3)
    outs.append(fn())
    o.gemm_ring(ring)
    return outs


bad = 0
for ci in range(cases):
    kind = rng.choice(["plain", "plain", "gate", "kseg", "conv", "gconv"])
    if kind in ("plain", "gate", "kseg"):
        M = rng.choice([1, 7, 32, 100, 129, 300, 1000, 2500, 4100, 9000, 256, 512, 1024, 2048, 4096, 6144])
        C = rng.choice([32, 64, 96, 128, 256, 384])
        nseg = rng.choice([1, 2, 3])
        act = rng.choice([o.ACT_NONE, o.ACT_RELU, o.ACT_LRELU])
        use_add = rng.random() < 0.5
        if kind == "plain":
            N, K = rng.choice([32, 64, 96, 128, 256, 512]), rng.choice([32, 64, 160, 256, 512, 1024])
            a, w, b = rnd(M, K).to(dev), rnd(N, K, scale=K ** -0.5).to(dev), rnd(N).to(dev)
            add = rnd(M, N).to(dev) if use_add else None
            ref = a.double() @ w.double().t() + b.double()
            ref = F.relu(ref) if act == o.ACT_RELU else (F.leaky_relu(ref, 0.01) if act == o.ACT_LRELU else ref)
            ref = ref + (add.double() if use_add else 0)

            def fn():
                out = torch.empty(M, N, device=dev)
                o.gemm(a, M, N, K, [w], out, biases=[b], act=act, slope=0.01, addend=add)
                return out
        elif kind == "gate":
            x = rnd(M, C).to(dev)
            wa = [rnd(C, C, scale=C ** -0.5).to(dev) for _ in range(nseg)]
            wb = [rnd(C, C, scale=C ** -0.5).to(dev) for _ in range(nseg)]
            ba = [rnd(C).to(dev) for _ in range(nseg)]
            bb = [rnd(C).to(dev) for _ in range(nseg)]
            xd = x.double()
            ref = torch.cat([(xd @ wa[i].double().t() + ba[i].double()) * F.relu(xd @ wb[i].double().t() + bb[i].double()) for i in range(nseg)], 1)

            def fn():
                out = torch.empty(M, nseg * C, device=dev)
                o.gemm(x, M, nseg * C, C, wa, out, weights2=wb, biases=ba, biases2=bb, act=o.ACT_GATE)
                return out
        else:
            hid = rnd(M, nseg * C).to(dev)
            wc = [rnd(C, C, scale=C ** -0.5).to(dev) for _ in range(nseg)]
            bc = [rnd(C).to(dev) for _ in range(nseg)]
            add = rnd(M, C).to(dev) if use_add else None
            ref = sum(hid[:, i * C:(i + 1) * C].double() @ wc[i].double().t() + bc[i].double() for i in range(nseg)) + (add.double() if use_add else 0)

            def fn():
                out = torch.empty(M, C, device=dev)
                o.gemm(hid, M, C, nseg * C, wc, out, biases=bc, seg_mode=o.SEG_K, addend=add)
                return out
    else:
        B, H, W = rng.choice([1, 2, 3, 5]), rng.choice([3, 4, 7, 8, 16, 33]), rng.choice([2, 4, 5, 8, 16, 40])
        M = B * H * W
        if kind == "conv":
            Cin, Cout = rng.choice([32, 64, 128]), rng.choice([32, 64, 128, 256])
            x = rnd(M, Cin).to(dev)
            w4 = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5)
            wk = w4.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().to(dev)
            b = rnd(Cout).to(dev)
            xn = x.cpu().double().reshape(B, H, W, Cin).permute(0, 3, 1, 2)
            ref = F.conv2d(xn, w4.double(), b.cpu().double(), padding=1).permute(0, 2, 3, 1).reshape(M, Cout).to(dev)

            def fn():
                out = torch.empty(M, Cout, device=dev)
                o.gemm(x, M, Cout, 9 * Cin, [wk], out, lda=Cin, ldw=9 * Cin, biases=[b], a_mode=o.A_CONV3X3, conv_hw=(H, W), cin=Cin)
                return out
        else:
            C = rng.choice([64, 128, 256])
            x = rnd(M, C).to(dev)
            wp = rnd(C, 288, scale=288 ** -0.5).to(dev)
            b = rnd(C).to(dev)
            add = rnd(M, C).to(dev)
            xn = x.cpu().double().reshape(B, H, W, C).permute(0, 3, 1, 2)
            ref = F.conv2d(xn, wp.cpu().double().reshape(C, 3, 3, 32).permute(0, 3, 1, 2), b.cpu().double(), padding=1, groups=C // 32)
            ref = (ref.permute(0, 2, 3, 1).reshape(M, C) + add.cpu().double()).to(dev)

            def fn():
                out = torch.zeros(M, C, device=dev)
                o.gemm(x, M, 32, 288, [wp], out, lda=C, ldw=288, biases=[b], addend=add, ldadd=C, ldo=C, a_mode=o.A_CONV3X3, conv_hw=(H, W),
                       cin=32, groups=C // 32, a_gstride=32, w_gstride=32 * 288, o_gstride=32, b_gstride=32)
                return out
    outs = run_all(fn)
    e = [rel(t, ref) for t in outs]
    ok = torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[3]) and max(e) < 1e-5
    if not ok:
        bad += 1
        print("MISMATCH case %d %s: errors %s, tile==stream %s, tile==ring %s" % (ci, kind, e, torch.equal(outs[0], outs[1]), torch.equal(outs[0], outs[3])), flush=True)
print("%d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
