"""Batch-1 latency of the reference's own usage (sample_ldm.py loops batch-1 samples): 50 steps + decode."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_image_generator_amd import synth
from ldm_image_generator_amd.ddpm import DDPM
from ldm_image_generator_amd.unet import UNet
from ldm_image_generator_amd.vae import Decoder
dev = torch.device("cuda:0")
net = UNet(); net.load_state_dict(synth.fill_state_dict(net.state_dict())); net = net.to(dev)
dec = Decoder(); dec.load_state_dict(synth.fill_state_dict(dec.state_dict())); dec = dec.to(dev)
d = DDPM(model=net)
for B in [int(v) for v in sys.argv[1:]] or (1, 4, 16):
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        z = d.sample((B, 8, 32, 32), seed=it, num_steps=50, progress=False)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        with torch.no_grad():
            img = dec(z)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print("B=%d: sample %.1f ms (%.2f ms/step), decode %.1f ms, %.2f images/s" % (B, (t1 - t0) * 1e3, (t1 - t0) * 20, (t2 - t1) * 1e3, B / (t2 - t0)), flush=True)
