"""Micro-benchmark of the GEMM family on the UNet / VAE shapes of BASELINE cfg 3 (B=256).
Usage: python tools/gemm_bench.py [variant ...]   (variants: 0 = tile, 1 = stream, 2 = stream + bf16x3 split consumer)"""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_image_generator_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=8):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def cases(B=256):
    out = []
    for s, (C, R) in enumerate([(128, 32), (256, 16), (512, 8), (1024, 4)]):
        M = B * R * R
        x = torch.randn(M, C, device=dev)
        wa = [torch.randn(C, C, device=dev) * C ** -0.5 for _ in range(3)]
        wb = [torch.randn(C, C, device=dev) * C ** -0.5 for _ in range(3)]
        wc = [torch.randn(C, C, device=dev) * C ** -0.5 for _ in range(3)]
        bs = [torch.randn(C, device=dev) for _ in range(3)]
        hid = torch.empty(M, 3 * C, device=dev)
        y = torch.randn(M, C, device=dev)
        out.append(("s%d gate  M=%d N=%d K=%d" % (s, M, 3 * C, C), 2.0 * M * 6 * C * C,
                    lambda x=x, wa=wa, wb=wb, bs=bs, hid=hid, M=M, C=C: ops.gemm(x, M, 3 * C, C, wa, hid, weights2=wb, biases=bs, biases2=bs, act=ops.ACT_GATE)))
        out.append(("s%d gemm2 M=%d N=%d K=%d" % (s, M, C, 3 * C), 2.0 * M * 3 * C * C,
                    lambda hid=hid, wc=wc, bs=bs, y=y, M=M, C=C: ops.gemm(hid, M, C, 3 * C, wc, y, biases=bs, seg_mode=ops.SEG_K, addend=y)))
        wq = torch.randn(3 * C, C, device=dev) * C ** -0.5
        bq = torch.randn(3 * C, device=dev)
        out.append(("s%d qkv   M=%d N=%d K=%d" % (s, M, 3 * C, C), 2.0 * M * 3 * C * C,
                    lambda x=x, wq=wq, bq=bq, hid=hid, M=M, C=C: ops.gemm(x, M, 3 * C, C, [wq], hid, biases=[bq])))
        wo = torch.randn(C, C, device=dev) * C ** -0.5
        out.append(("s%d oproj M=%d N=%d K=%d" % (s, M, C, C), 2.0 * M * C * C,
                    lambda x=x, wo=wo, bs=bs, y=y, M=M, C=C: ops.gemm(x, M, C, C, [wo], y, biases=[bs[0]], addend=y)))
        wg = torch.randn(C, 288, device=dev) * 288 ** -0.5
        out.append(("s%d gconv M=%d C=%d" % (s, M, C), 2.0 * M * C * 288,
                    lambda x=x, wg=wg, bs=bs, y=y, M=M, C=C, R=R: ops.gemm(x, M, 32, 288, [wg], y, lda=C, ldw=288, biases=[bs[0]], addend=x, ldadd=C, ldo=C,
                                                                            a_mode=ops.A_CONV3X3, conv_hw=(R, R), cin=32, groups=C // 32, a_gstride=32,
                                                                            w_gstride=32 * 288, o_gstride=32, b_gstride=32)))
    for (C, R, Bv) in [(512, 32, 64), (256, 64, 64), (128, 128, 64), (64, 256, 32)]:
        M = Bv * R * R
        x = torch.randn(M, C, device=dev)
        w = torch.randn(C, 9 * C, device=dev) * (9 * C) ** -0.5
        b = torch.randn(C, device=dev)
        y = torch.empty(M, C, device=dev)
        out.append(("vae conv3x3 C=%d R=%d B=%d" % (C, R, Bv), 2.0 * M * C * 9 * C,
                    lambda x=x, w=w, b=b, y=y, M=M, C=C, R=R: ops.gemm(x, M, C, 9 * C, [w], y, lda=C, ldw=9 * C, biases=[b], act=ops.ACT_LRELU, slope=0.01,
                                                                        addend=x, a_mode=ops.A_CONV3X3, conv_hw=(R, R), cin=C)))
    return out


if __name__ == "__main__":
    variants = [int(v) for v in sys.argv[1:]] or [0, 1]
    if os.environ.get("GEMM_WIDE") is not None:
        ops.gemm_wide_epilogue(int(os.environ["GEMM_WIDE"]))
    cs = cases(int(os.environ.get('GEMM_BATCH', '256')))
    flt = os.environ.get('GEMM_CASES')
    if flt:
        cs = [c for c in cs if any(f in c[0] for f in flt.split(','))]
    print("%-44s" % "case" + "".join("   v%d ms   TF/s" % v for v in variants))
    for name, flops, fn in cs:
        row = "%-44s" % name
        for v in variants:
            old = ops.gemm_variant(v)
            ms = timeit(fn)
            ops.gemm_variant(old)
            row += " %8.3f %6.1f" % (ms, flops / ms / 1e9)
        print(row, flush=True)
