"""Per-shape census of the MFMA launches of one UNet forward (native executor) at the headline shape, bf16 or fp32 mode:
`python3 tools/census16.py [--fp32] [--batch B] [--latent R]`.  Groups the profiler's records by (class, FLOPs, bytes) = by layer shape."""
import argparse
import collections
import ctypes
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_image_generator_amd import _lib, autocast, ops, synth          # noqa: E402
from ldm_image_generator_amd.unet import UNet                          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--fp32", action="store_true")
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--latent", type=int, default=32)
args = ap.parse_args()
dev = torch.device("cuda:0")
net = UNet()
net.load_state_dict(synth.fill_state_dict(net.state_dict()))
net = net.to(dev).eval()
if not args.fp32:
    autocast.set_autocast_dtype(net, torch.bfloat16)
    net._autocast_now = True
B, R = args.batch, args.latent
x = torch.randn(B, 8, R, R, device=dev)
t = torch.full((B,), 500, device=dev)
names = {0: "f32", 3: "bf16", 5: "gconv16"}
with torch.no_grad():
    random.seed(0)
    net(x=x, time=t, condition=None)
    torch.cuda.synchronize()
    ops.prof_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    random.seed(0)
    net(x=x, time=t, condition=None)
    e1.record()
    torch.cuda.synchronize()
buf = (ctypes.c_double * (4 * 4096))()
n = _lib.load().ldm_prof_dump(buf, 4096)
agg = collections.OrderedDict()
for i in range(n):
    key = (int(buf[4 * i]), buf[4 * i + 2], buf[4 * i + 3])
    c, ms = agg.get(key, (0, 0.0))
    agg[key] = (c + 1, ms + buf[4 * i + 1])
tot = sum(v[1] for v in agg.values())
print("forward %.3f ms wall (with events); %d MFMA launches, %.3f ms in them" % (e0.elapsed_time(e1), n, tot))
print("%-8s %5s %10s %9s %9s %9s %9s" % ("class", "calls", "GFLOP", "MB", "us each", "TFLOP/s", "GB/s"))
for (cls, fl, by), (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-8s %5d %10.2f %9.1f %9.1f %9.1f %9.0f   total %.3f ms" % (names.get(cls, str(cls)), c, fl / 1e9, by / 1e6, ms / c * 1e3, fl * c / ms / 1e9,
                                                                   by * c / ms / 1e6, ms))
ops.prof_enable(False)
