"""Per-shape census of the MFMA launches of one training step (cfg 5 per-GPU shape): `python3 tools/census_train.py [--precision bf16]`."""
import argparse
import collections
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_image_generator_amd import _lib, dist as ldist, ops, synth, train as ltrain  # noqa: E402
from ldm_image_generator_amd.ddpm import DDPM  # noqa: E402
from ldm_image_generator_amd.unet import UNet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16")
ap.add_argument("--batch", type=int, default=128)
args = ap.parse_args()
dev = torch.device("cuda:0")
net = UNet()
net.load_state_dict(synth.fill_state_dict(net.state_dict()))
net = net.to(dev).train()
ltrain.set_precision(net, args.precision)
ddpm = DDPM(model=net)
opt = torch.optim.AdamW(ddpm.parameters(), lr=1e-4, fused=True)
x = torch.randn(args.batch, 8, 64, 64, generator=torch.Generator().manual_seed(0)).to(dev)
for i in range(2):
    ldist.train_step(ddpm, opt, x, 0, 1)
torch.cuda.synchronize()
ops.prof_enable(True)
ldist.train_step(ddpm, opt, x, 1, 1)
torch.cuda.synchronize()
buf = (ctypes.c_double * (4 * 8192))()
n = _lib.load().ldm_prof_dump(buf, 8192)
names = {0: "f32", 1: "tn_f32", 2: "gcwg_f32", 3: "bf16", 4: "tn_bf16", 5: "gconv16"}
agg = collections.OrderedDict()
for i in range(n):
    key = (int(buf[4 * i]), buf[4 * i + 2], buf[4 * i + 3])
    c, ms = agg.get(key, (0, 0.0))
    agg[key] = (c + 1, ms + buf[4 * i + 1])
tot = sum(v[1] for v in agg.values())
print("%d MFMA launches, %.2f ms in them" % (n, tot))
print("%-8s %5s %10s %9s %9s %9s %9s" % ("class", "calls", "GFLOP", "MB", "us each", "TFLOP/s", "GB/s"))
for (cls, fl, by), (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%-8s %5d %10.2f %9.1f %9.1f %9.1f %9.0f   total %.3f ms" % (names.get(cls, str(cls)), c, fl / 1e9, by / 1e6, ms / c * 1e3, fl * c / ms / 1e9,
                                                                   by * c / ms / 1e6, ms))
ops.prof_enable(False)
