import sys, os, torch, statistics
sys.path.insert(0, os.getcwd())
from ldm_image_generator_amd import ops
dev = torch.device("cuda:0")
for (M, N, K, add) in [(262144, 128, 384, True), (262144, 128, 384, False), (65536, 256, 768, True), (262144, 384, 128, False)]:
    a = torch.randn(M, K, device=dev); out = torch.zeros(M, N, device=dev)
    nseg = 3 if add or K > N else 1
    w = [torch.randn(N, K // 3, device=dev) * K ** -0.5 for _ in range(3)] if K > N else [torch.randn(N, K, device=dev) * K ** -0.5]
    st = torch.zeros(1024, dtype=torch.int64, device=dev)
    ops.gemm_ring(2)
    for _ in range(3):
        st.zero_()
        if K > N:
            ops.gemm(a, M, N, K, w, out, seg_mode=ops.SEG_K, addend=out if add else None, biases2=[st.view(torch.float32)] + [None] * (len(w) - 1))
        else:
            ops.gemm(a, M, N, K, w, out, biases2=[st.view(torch.float32)] + [None] * (len(w) - 1))
        torch.cuda.synchronize()
    s = st.cpu().tolist()
    kl = [s[c*4+1]-s[c*4+0] for c in range(1, 4)]; ep = [s[c*4+2]-s[c*4+1] for c in range(1, 4)]; gap = [s[(c+1)*4+0]-s[c*4+2] for c in range(1, 3)]
    clk = (s[202]-s[200]) / ((s[203]-s[201]) * 10.0) if s[203] != s[201] else 0
    print("M=%d N=%d K=%d add=%s: K-loop %s cycles, epilogue %s, next-start gap %s; clock %.2f GHz; steps of tile 1: %s" % (M, N, K, add, kl, ep, gap, clk, [s[64+k+1]-s[64+k] for k in range(6)]))
