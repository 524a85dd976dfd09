"""BASELINE cfg 2: sample_ddpm.py shape -- UNet(input_channels=3) on 64x64 pixels, 50 DDIM steps, batch 64, no VAE.
    python tools/cfg2_bench.py [--batch 64] [--gemm-variant 1]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_image_generator_amd import ops, synth  # noqa: E402
from ldm_image_generator_amd.ddpm import DDPM  # noqa: E402
from ldm_image_generator_amd.unet import UNet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--gemm-variant", type=int, default=1)
args = ap.parse_args()
dev = torch.device("cuda:0")
ops.gemm_variant(args.gemm_variant)
net = UNet(input_channels=3)
net.load_state_dict(synth.fill_state_dict(net.state_dict()))
net = net.to(dev).eval()
d = DDPM(model=net)
x_t = torch.randn(args.batch, 3, 64, 64, generator=torch.Generator().manual_seed(0)).to(dev)
d.sample(tuple(x_t.shape), seed=0, num_steps=50, x_init=x_t, progress=False)
torch.cuda.synchronize()
ops.prof_enable(True)
t0 = time.perf_counter()
out = d.sample(tuple(x_t.shape), seed=1, num_steps=50, x_init=x_t, progress=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n, ms, fl = ops.prof_read()
ops.prof_enable(False)
print("cfg2 batch %d: %.1f ms per 50-step sample call, %.1f denoise-steps/s, %.0f sample-steps/s, %.1f images/s | GEMM family %.1f TFLOP/s, finite %s"
      % (args.batch, dt * 1e3, 50 / dt, 50 * args.batch / dt, args.batch / dt, fl / ms / 1e9, bool(torch.isfinite(out).all())))
