import sys, os, torch
sys.path.insert(0, os.getcwd())
from ldm_image_generator_amd import ops
dev = torch.device("cuda:0")
def timed(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for (B, R, C) in [(256, 32, 128), (256, 16, 256), (256, 8, 512), (256, 4, 1024)]:
    HW = R * R
    torch.manual_seed(0); x = torch.randn(B * HW, C, device=dev); film = torch.randn(HW, 2 * C, device=dev); out = torch.empty_like(x)
    us = timed(lambda: ops.channelnorm_film(x, film, None, out, B, HW, C))
    print("channelnorm_film B=%d R=%d C=%d: %.1f us, %.2f TB/s, bits %d" % (B, R, C, us, 2 * x.numel() * 4 / us / 1e6, int(out.view(torch.int32).long().sum())))
