"""Steady-state probes: (a) L2-resident operands, long K; (b) HBM-streaming A."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_image_generator_amd import ops
from gemm_bench import timeit
dev = torch.device("cuda:0")
for (M, N, K) in [(4096, 4096, 4096), (8192, 2048, 8192), (65536, 1024, 1024), (262144, 1024, 256), (16384, 16384, 512)]:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; out = torch.empty(M, N, device=dev)
    row = "M=%d N=%d K=%d " % (M, N, K)
    for v in (0, 1):
        old = ops.gemm_variant(v)
        ms = timeit(lambda: ops.gemm(a, M, N, K, [w], out))
        ops.gemm_variant(old)
        row += "  v%d %.3f ms %.1f TF/s" % (v, ms, 2.0 * M * N * K / ms / 1e9)
    print(row, flush=True)
