"""UNet training-step throughput on the HIP path (BASELINE cfg 5 shape per GPU: latents [B, 8, 64, 64]).
    python tools/train_bench.py [--batch 128] [--latent 64] [--steps 3]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_image_generator_amd import dist as ldist, ops, synth  # noqa: E402
from ldm_image_generator_amd.ddpm import DDPM  # noqa: E402
from ldm_image_generator_amd.unet import UNet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--latent", type=int, default=64)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2, help="untimed steps before the timed ones (the first two grow the caching allocator)")
ap.add_argument("--gemm-variant", type=int, default=1, help="1 = exact-fp32 stream schedule, 2 = bf16x3 split consumer")
ap.add_argument("--fused-adamw", type=int, default=1, help="1 (default, as bench.py): torch.optim.AdamW(fused=True); 0: the foreach default")
ap.add_argument("--precision", default="f32", choices=["f32", "bf16"], help="operand precision of the training step (train.set_precision)")
args = ap.parse_args()
dev = torch.device("cuda:0")
ops.gemm_variant(args.gemm_variant)
net = UNet()
net.load_state_dict(synth.fill_state_dict(net.state_dict()))
net = net.to(dev).train()
from ldm_image_generator_amd import train as ltrain  # noqa: E402
ltrain.set_precision(net, args.precision)
ddpm = DDPM(model=net)
opt = torch.optim.AdamW(ddpm.parameters(), lr=1e-4, fused=bool(args.fused_adamw))
x = torch.randn(args.batch, 8, args.latent, args.latent, generator=torch.Generator().manual_seed(0)).to(dev)
for i in range(args.warmup):
    ldist.train_step(ddpm, opt, x, 0, 1)
torch.cuda.synchronize()
ops.prof_enable(True)
t0 = time.perf_counter()
for i in range(args.steps):
    loss = ldist.train_step(ddpm, opt, x, 1 + i, 1)
host = (time.perf_counter() - t0) / args.steps          # time the host needed to ENQUEUE a step (no synchronisation inside a step)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
print("host enqueue %.1f ms/step of %.1f ms/step" % (host * 1e3, dt * 1e3))
n, ms, fl = ops.prof_read()
ops.prof_enable(False)
print("batch %d latent %d: %.1f ms/step, %.1f samples/s, loss %.4f | GEMM launches/step %d, GEMM %.1f ms/step at %.1f TFLOP/s, peak mem %.1f GB"
      % (args.batch, args.latent, dt * 1e3, args.batch / dt, float(loss), n // args.steps, ms / args.steps, fl / ms / 1e9,
         torch.cuda.max_memory_allocated() / 2 ** 30))
