"""Per-shape census of the MFMA launches of one train_vae.py iteration (256x256, batch 8): `python3 tools/census_vae.py`."""
import collections
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_image_generator_amd import _lib, ops, synth  # noqa: E402
from ldm_image_generator_amd.vae import VAE, Decoder, Discriminator, Encoder, VectorQuantizer  # noqa: E402

dev = torch.device("cuda:0")
enc, dec, disc = Encoder(), Decoder(), Discriminator()
for m in (enc, dec, disc):
    m.load_state_dict(synth.fill_state_dict(m.state_dict()))
torch.manual_seed(1234)
vae = VAE(enc, dec, VectorQuantizer()).to(dev)
disc = disc.to(dev)
opt_v = torch.optim.AdamW(vae.parameters(), lr=1e-4, fused=True)
opt_d = torch.optim.AdamW(disc.parameters(), lr=1e-4, fused=True)
img = (torch.rand(8, 3, 256, 256, generator=torch.Generator().manual_seed(7)) * 2 - 1).to(dev)


def it():
    opt_v.zero_grad()
    recon, reg, y = vae.calclate_loss(img)
    adv = torch.relu(-disc.calclate_logit(y)).mean()
    (recon * 10.0 + reg * 1.0 + adv * 0.1).backward()
    opt_v.step()
    opt_d.zero_grad()
    y = y.detach()
    d_loss = torch.relu(1 + disc.calclate_logit(y)).mean() + torch.relu(1 - disc.calclate_logit(img)).mean()
    d_loss.backward()
    opt_d.step()


for _ in range(2):
    it()
torch.cuda.synchronize()
ops.prof_enable(True)
it()
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
for (cls, fl, by), (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-8s %5d %10.2f %9.1f %9.1f %9.1f %9.0f   total %.3f ms" % (names.get(cls, str(cls)), c, fl / 1e9, by / 1e6, ms / c * 1e3, fl * c / ms / 1e9,
                                                                   by * c / ms / 1e6 if by else 0.0, ms))
ops.prof_enable(False)
