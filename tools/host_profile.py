"""Where the HOST time of a training step goes: UNetFunction.forward / .backward called directly (main thread) under cProfile.
    python3 tools/host_profile.py [--precision bf16]"""
import argparse
import cProfile
import os
import pstats
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_image_generator_amd import synth, train as ltrain  # noqa: E402
from ldm_image_generator_amd.unet import UNet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16")
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--top", type=int, default=40)
args = ap.parse_args()
dev = torch.device("cuda:0")
net = UNet()
net.load_state_dict(synth.fill_state_dict(net.state_dict()))
net = net.to(dev).train()
ltrain.set_precision(net, args.precision)
params = [p for p in net.parameters() if p.requires_grad]
x = torch.randn(args.batch, 8, 64, 64, device=dev)
t = torch.randint(1, 1000, (args.batch,), device=dev)


class Ctx:
    needs_input_grad = (False, False, False)


def step(prof=None):
    ctx = Ctx()
    random.seed(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if prof:
        prof.enable()
    with torch.no_grad():
        out = ltrain.UNetFunction.forward(ctx, net, x, t, *params)
        t1 = time.perf_counter()
        ltrain.UNetFunction.backward(ctx, torch.ones_like(out))
    if prof:
        prof.disable()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t0) * 1e3


for _ in range(2):
    step()
print("host forward %.1f ms, host backward %.1f ms, wall incl. GPU drain %.1f ms" % step())
pr = cProfile.Profile()
step(pr)
pstats.Stats(pr).sort_stats("tottime").print_stats(args.top)
