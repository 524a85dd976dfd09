"""How much of a small-batch denoise step is launch overhead?  One full-size UNet forward, eager vs replayed from a
HIP graph (decisions fixed at capture -- a probe, not a product path).   python tools/graph_probe.py [batch ...]"""
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_image_generator_amd import synth  # noqa: E402
from ldm_image_generator_amd.unet import UNet  # noqa: E402

dev = torch.device("cuda:0")
net = UNet()
net.load_state_dict(synth.fill_state_dict(net.state_dict()))
net = net.to(dev).eval()
for B in [int(v) for v in sys.argv[1:]] or (1, 4, 16):
    x = torch.randn(B, 8, 32, 32, device=dev)
    t1 = torch.tensor([500], dtype=torch.int64, device=dev)
    tt = torch.full((B,), 500, device=dev)

    def fwd():
        random.seed(3)
        net._uniform_time = (500, t1)
        try:
            return net(x, tt)
        finally:
            net._uniform_time = None

    with torch.no_grad():
        fwd()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fwd()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = fwd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fwd()
        torch.cuda.synchronize()
        te = (time.perf_counter() - t0) / 20
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        tg = (time.perf_counter() - t0) / 20
    print("B=%d: UNet forward eager %.2f ms, graph replay %.2f ms" % (B, te * 1e3, tg * 1e3), flush=True)
