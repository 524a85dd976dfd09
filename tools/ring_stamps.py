"""In-kernel cycle stamps of the ring GEMM kernel (probe build): where a tile's cycles go, the shader clock the chip sustains under
the exact-fp32 MFMA stream, and the spread of the workgroups' run times.

    hipcc --offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -ffp-contract=off -DLDM_RING_STAMP=1 \
          -o /tmp/libldm_stamp.so ldm_image_generator_amd/csrc/*.hip ldm_image_generator_amd/csrc/*.cpp
    LDM_HIP_LIB=/tmp/libldm_stamp.so python tools/ring_stamps.py

The probe build writes s_memtime / s_memrealtime stamps of workgroup 0 / wave 0 (and one realtime pair per workgroup) into the buffer
passed as bias2[0] of a plain problem; the product build contains none of it."""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from ldm_image_generator_amd import ops
dev = torch.device("cuda:0")
M, N, K = 262144, 768, 128
a = torch.randn(M, K, device=dev); out = torch.zeros(M, N, device=dev)
w = [torch.randn(N, K, device=dev) * K ** -0.5]
st = torch.zeros(1024, dtype=torch.int64, device=dev)
ops.gemm_ring(2)
for _ in range(3):
    st.zero_()
    ops.gemm(a, M, N, K, w, out, biases2=[st.view(torch.float32)])
    torch.cuda.synchronize()
s = st.cpu().tolist()
print("tile: start->Kloop_end  Kloop_end->epi_end  epi_end->next_start")
for c in range(11):
    print(c, s[c*4+1]-s[c*4+0], s[c*4+2]-s[c*4+1], s[(c+1)*4+0]-s[c*4+2])
print("in-kernel clock over tiles 0..10: %.3f GHz (d_memtime %d, d_realtime %d x 10 ns)" % ((s[202]-s[200]) / ((s[203]-s[201]) * 10.0), s[202]-s[200], s[203]-s[201]))
starts = [s[256 + 2 * b] for b in range(256)]; ends = [s[257 + 2 * b] for b in range(256)]
t0 = min(starts)
import statistics
print("workgroup starts (us after first): min %.1f median %.1f max %.1f | ends: min %.1f median %.1f max %.1f | durations: min %.1f median %.1f max %.1f" % (
    0.0, statistics.median(x - t0 for x in starts) / 100, (max(starts) - t0) / 100, (min(ends) - t0) / 100, statistics.median(x - t0 for x in ends) / 100, (max(ends) - t0) / 100,
    min(e - b for b, e in zip(starts, ends)) / 100, statistics.median(e - b for b, e in zip(starts, ends)) / 100, max(e - b for b, e in zip(starts, ends)) / 100))
by_xcd = [statistics.median((ends[b] - starts[b]) / 100 for b in range(x, 256, 8)) for x in range(8)]
print("median duration per XCD label (us):", ["%.0f" % v for v in by_xcd])
print("tile 1 steps (cycles between step ends):", [s[64+k+1]-s[64+k] for k in range(7)], "first step end - tile start:", s[64]-s[4])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for mode in (2, 0):
    ops.gemm_ring(mode)
    ops.gemm(a, M, N, K, w, out); torch.cuda.synchronize()
    e0.record()
    for _ in range(10): ops.gemm(a, M, N, K, w, out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print("mode", mode, "%.1f us  %.1f TF" % (us, 2.0 * M * N * K / us / 1e6))
