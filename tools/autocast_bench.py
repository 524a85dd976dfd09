"""bf16 (autocast) sampling + decode alone: `python3 tools/autocast_bench.py [--passes N] [--batch B] [--no-decode] [--fp32]`.
Used under rocprofv3 --kernel-trace --stats for the per-kernel picture of the opt-in mode (bench.py's `autocast_bf16` leg)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passes", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--num-steps", type=int, default=50)
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--fp32", action="store_true")
    args = ap.parse_args()
    from ldm_image_generator_amd import autocast, synth
    from ldm_image_generator_amd.ddpm import DDPM
    from ldm_image_generator_amd.unet import UNet
    from ldm_image_generator_amd.vae import Decoder
    dev = torch.device("cuda", 0)
    net = UNet()
    net.load_state_dict(synth.fill_state_dict(net.state_dict()))
    dec = Decoder()
    dec.load_state_dict(synth.fill_state_dict(dec.state_dict()))
    net, dec = net.to(dev).eval(), dec.to(dev)
    if not args.fp32:
        autocast.set_autocast_dtype(net, torch.bfloat16)
        autocast.set_compute_dtype(dec, torch.bfloat16)
    d = DDPM(model=net)
    B = args.batch
    x_t = torch.randn(B, 8, 32, 32, generator=torch.Generator().manual_seed(0)).to(dev)

    def one(seed):
        z = d.sample((B, 8, 32, 32), seed=seed, num_steps=args.num_steps, x_init=x_t, progress=False)
        if args.no_decode:
            return z
        with torch.no_grad():
            return dec(z)

    one(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.passes):
        out = one(1 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.passes
    print("%s: %.1f ms per pass, %.1f images/s, finite=%s" % ("fp32" if args.fp32 else "bf16", dt * 1e3, B / dt, bool(torch.isfinite(out).all())))


if __name__ == "__main__":
    main()
