"""Generate the golden fixtures in this directory by IMPORTING the reference.

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Every fixture is data: formula-generated weights (``synth.fill_state_dict``; the
weights themselves are NOT stored, they are regenerated from the key names),
inputs, and the outputs the reference's own modules produced on CPU (torch
2.10.0+rocm7.0, fp32).  Nothing of the reference's source is copied.
"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

import attention as ref_attention      # noqa: E402
import modules as ref_modules          # noqa: E402
import sinusoidal as ref_sin           # noqa: E402
import unet as ref_unet                # noqa: E402
import vae as ref_vae                  # noqa: E402
import ddpm as ref_ddpm                # noqa: E402
from ldm_image_generator_amd import synth  # noqa: E402

torch.set_num_threads(8)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def load_formula(module, salt=0, gain=1.0):
    module.load_state_dict(synth.fill_state_dict(module.state_dict(), salt=salt, gain=gain))
    return module


class Trace:
    """Record the reference's Python-RNG decisions without touching its code."""

    def __enter__(self):
        self.events = []
        self._r, self._s = random.random, random.sample

        def rnd():
            v = self._r()
            self.events.append(("r", v))
            return v

        def smp(pop, k):
            idx = self._s(range(len(pop)), k)
            self.events.append(("s", tuple(idx)))
            return [pop[i] for i in idx]

        random.random, random.sample = rnd, smp
        return self

    def __exit__(self, *a):
        random.random, random.sample = self._r, self._s

    def encoded(self):
        # one row per event: [kind(0=random,1=sample), value or e1, e2]
        rows = []
        for e in self.events:
            if e[0] == "r":
                rows.append([0.0, e[1], -1.0])
            else:
                rows.append([1.0, float(e[1][0]), float(e[1][1])])
        return np.asarray(rows, dtype=np.float64).reshape(-1, 3)


def g(key, shape, salt=0):
    return synth.gaussian_tensor(key, shape, salt)


@torch.no_grad()
def tables():
    arrs = {}
    for c, h, w in [(32, 7, 5), (128, 32, 32), (1024, 4, 4)]:
        x = torch.zeros(1, c, h, w)
        pe = ref_sin.PositionalEncoding2d(c, return_encoding_only=True)(x)
        arrs["pe_%d_%d_%d" % (c, h, w)] = pe[0]
    steps = torch.linspace(0, 999, 50).int()
    for c in (32, 128, 1024):
        x = torch.zeros(50, c, 1, 1)
        te = ref_sin.TimeEncoding2d(c, return_encoding_only=True)(x, steps.long())
        arrs["te_%d" % c] = te[:, :, 0, 0]
    arrs["te_steps"] = steps
    save("tables", **arrs)


@torch.no_grad()
def schedule():
    d = ref_ddpm.DDPM(model=torch.nn.Conv2d(1, 1, 1))
    arrs = dict(beta=d.beta, alpha_bar=d.alpha_bar, alpha_cum=torch.cumprod(1 - d.beta, dim=0))
    for n in (5, 20, 50):
        arrs["steps_%d" % n] = torch.linspace(0, 999, n).int()
    save("schedule", **arrs)


@torch.no_grad()
def module_cases():
    # ChannelNorm
    x = g("cn.x", (2, 32, 5, 7)) * 3 + 0.5
    save("channel_norm", x=x, y=ref_modules.ChannelNorm(32)(x))
    # Encodings
    enc = load_formula(ref_unet.Encodings(32))
    x = g("enc.x", (3, 32, 6, 5))
    t = torch.tensor([3, 500, 3])
    save("encodings", x=x, t=t, y=enc(x, t))
    # ReGLU / RandomMoE
    moe = load_formula(ref_modules.RandomMoE(32))
    x = g("moe.x", (2, 32, 4, 6))
    outs = {}
    for seed in (0, 1, 7):
        random.seed(seed)
        with Trace() as tr:
            y = moe(x)
        outs["y_%d" % seed] = y
        outs["picks_%d" % seed] = np.asarray(tr.events[0][1])
    save("random_moe", x=x, general=moe.general(x), **outs)
    # grouped conv (unet.py:30)
    conv = load_formula(torch.nn.Conv2d(64, 64, 3, 1, 1, groups=2))
    x = g("gconv.x", (2, 64, 6, 9))
    save("gconv", x=x, y=conv(x))
    # WindowAttention: every pad / shift / global case
    arrs = {}
    for c in (64,):
        for shift in (0, 3):
            wa = load_formula(ref_attention.WindowAttention(c, n_heads=c // 32, window_size=6, shift=shift), gain=2.0)
            wa.train()
            for (h, w) in [(8, 8), (16, 16), (12, 12), (7, 9), (4, 4), (6, 6), (32, 32)]:
                x = g("wa.x.%d.%d" % (h, w), (2, c, h, w))
                arrs["x_%d_%d" % (h, w)] = x
                arrs["y_s%d_%d_%d" % (shift, h, w)] = wa(x)
    save("window_attention", **arrs)
    # SwinBlock, eval + train
    for attn, shift in ((True, 3), (True, 0), (False, 0)):
        blk = load_formula(ref_unet.SwinBlock(64, shift=shift, attention=attn))
        x = g("blk.x", (2, 64, 8, 8))
        t = torch.tensor([999, 20])
        arrs = dict(x=x, t=t)
        blk.eval()
        random.seed(5)
        with Trace() as tr:
            arrs["y_eval"] = blk(x, t)
        arrs["trace_eval"] = tr.encoded()
        blk.train()
        for seed in (0, 3):
            random.seed(seed)
            with Trace() as tr:
                arrs["y_train_%d" % seed] = blk(x, t)
            arrs["trace_train_%d" % seed] = tr.encoded()
        save("swin_block_a%d_s%d" % (int(attn), shift), **arrs)


TINY = dict(input_channels=8, stages=[1, 2, 3, 2], channels=[32, 64, 96, 128])


@torch.no_grad()
def unet_cases():
    net = load_formula(ref_unet.UNet(**TINY))
    x = g("tiny.x", (2, 8, 32, 32))
    t = torch.tensor([978, 40])
    arrs = dict(x=x, t=t)
    net.eval()
    random.seed(11)
    with Trace() as tr:
        arrs["y_eval"] = net(x, t)
    arrs["trace_eval"] = tr.encoded()
    net.train()
    for seed in (0, 1):
        random.seed(seed)
        with Trace() as tr:
            arrs["y_train_%d" % seed] = net(x, t)
        arrs["trace_train_%d" % seed] = tr.encoded()
    # pixel-space variant (BASELINE cfg 1/2: input_channels=3), non-multiple-of-6 sizes
    save("unet_tiny", **arrs)
    net3 = load_formula(ref_unet.UNet(input_channels=3, stages=[1, 2], channels=[32, 64]))
    x3 = g("tiny3.x", (3, 3, 16, 16))
    t3 = torch.tensor([5, 5, 700])
    net3.eval()
    random.seed(2)
    save("unet_tiny3", x=x3, t=t3, y_eval=net3(x3, t3))

    # full-size UNet (385.7 M parameters), formula weights
    full = load_formula(ref_unet.UNet())
    x = g("full.x", (2, 8, 32, 32))
    t = torch.tensor([999, 489])
    arrs = dict(x=x, t=t)
    full.eval()
    random.seed(0)
    arrs["y_eval"] = full(x, t)
    full.train()
    random.seed(0)
    with Trace() as tr:
        arrs["y_train_0"] = full(x, t)
    arrs["trace_train_0"] = tr.encoded()
    save("unet_full", **arrs)
    return net, full


@torch.no_grad()
def sample_cases(tiny, full):
    for name, net, kw in (("tiny", tiny, {}), ("full", full, {})):
        d = ref_ddpm.DDPM(model=net)
        arrs = {}
        for mode in ("train", "eval"):
            net.train(mode == "train")
            for steps in ((5, 50) if name == "tiny" else (3, 50)):
                if name == "full" and mode == "eval" and steps == 50:
                    continue
                torch.manual_seed(0)
                x_t = torch.randn(2 if name == "tiny" else 1, 8, 32, 32)
                arrs["xT"] = x_t
                with Trace() as tr:
                    y = d.sample(tuple(x_t.shape), seed=0, num_steps=steps, use_autocast=False)
                arrs["x0_%s_%d" % (mode, steps)] = y
                arrs["trace_%s_%d" % (mode, steps)] = tr.encoded()
        save("sample_" + name, **arrs)


def loss_cases(tiny):
    tiny.train()
    d = ref_ddpm.DDPM(model=tiny)
    x = g("loss.x", (4, 8, 32, 32))
    torch.manual_seed(3)
    random.seed(3)
    # replay what calculate_loss will draw (ddpm.py:40,44) so the fixture can hold it
    st = torch.get_rng_state()
    t = torch.randint(low=1, high=1000, size=(4,))
    e = torch.randn(4, 8, 32, 32)
    torch.set_rng_state(st)
    for p in tiny.parameters():
        p.grad = None
    with Trace() as tr:
        loss = d.calculate_loss(x)
    loss.backward()
    arrs = dict(x=x, t=t, e=e, loss=loss.detach(), trace=tr.encoded())
    names, norms = [], []
    for k, p in tiny.named_parameters():
        names.append(k)
        norms.append(-1.0 if p.grad is None else float(p.grad.double().norm()))
    arrs["grad_names"] = np.asarray(names)
    arrs["grad_norms"] = np.asarray(norms)
    arrs["grad_encoder_first_weight"] = tiny.encoder_first.weight.grad
    arrs["grad_decoder_last_weight"] = tiny.decoder_last.weight.grad
    save("loss_tiny", **arrs)
    for p in tiny.parameters():
        p.grad = None


def loss_full_cases():
    """Training parity at FULL width (cfg 5): default UNet() (385.7 M parameters, C = 128..1024), formula weights,
    latents [2, 8, 64, 64] (121-window attention at stage 0) and [2, 8, 32, 32]; fixed t / e / Python-random seed.
    Stored: loss, every per-parameter gradient norm (-1 where autograd leaves None) and a few small gradient tensors."""
    full = load_formula(ref_unet.UNet())
    full.train()
    d = ref_ddpm.DDPM(model=full)
    arrs = {}
    for tag, res, seed in (("r64", 64, 5), ("r32", 32, 6)):
        x = g("lossfull.x.%d" % res, (2, 8, res, res))
        torch.manual_seed(seed)
        random.seed(seed)
        st = torch.get_rng_state()
        t = torch.randint(low=1, high=1000, size=(2,))
        e = torch.randn(2, 8, res, res)
        torch.set_rng_state(st)
        for p in full.parameters():
            p.grad = None
        with Trace() as tr:
            loss = d.calculate_loss(x)
        loss.backward()
        arrs.update({"x_" + tag: x, "t_" + tag: t, "e_" + tag: e, "loss_" + tag: loss.detach(), "trace_" + tag: tr.encoded()})
        names, norms = [], []
        sd_grads = {}
        for k, p in full.named_parameters():
            names.append(k)
            norms.append(-1.0 if p.grad is None else float(p.grad.double().norm()))
            sd_grads[k] = p.grad
        arrs["grad_names"] = np.asarray(names)
        arrs["grad_norms_" + tag] = np.asarray(norms)
        arrs["grad_encoder_first_weight_" + tag] = full.encoder_first.weight.grad
        arrs["grad_decoder_last_weight_" + tag] = full.decoder_last.weight.grad
        # small slices of large gradients that exist in this draw (first executed block of each kind)
        def first(pred):
            for k in names:
                if pred(k) and sd_grads[k] is not None:
                    return k
            return None
        picks = {
            "expert_c1024": first(lambda k: k.startswith("encoder_stages.3") and ".ffn.experts." in k and k.endswith(".c.weight")),
            "general_a128": first(lambda k: k.startswith("decoder_stages.3") and ".ffn.general.a.weight" in k),
            "in_proj": first(lambda k: k.endswith("self_attention.attention.in_proj_weight") and k.startswith("decoder_stages.3")),
            "in_proj_bias": first(lambda k: k.endswith("self_attention.attention.in_proj_bias") and k.startswith("decoder_stages.2")),
            "gconv512": first(lambda k: k.startswith("encoder_stages.2") and k.endswith(".conv.weight")),
            "gconv128": first(lambda k: k.startswith("decoder_stages.3") and k.endswith(".conv.weight")),
            "proj1": first(lambda k: k.startswith("encoder_stages.0") and k.endswith("encodings.proj1.weight")),
            "down1": "encoder_stages.1.ch_conv.0.weight",
            "up2": "decoder_stages.2.ch_conv.1.weight",
        }
        for nick, k in picks.items():
            if k is None:
                continue
            gk = sd_grads[k]
            sl = gk.reshape(gk.shape[0], -1)[:64, :96] if gk.ndim > 1 else gk[:256]
            arrs["gslice_%s_%s" % (nick, tag)] = sl
            arrs["gslice_%s_%s_key" % (nick, tag)] = np.asarray(k)
        for p in full.parameters():
            p.grad = None
    save("loss_full", **arrs)


@torch.no_grad()
def vae_cases():
    rb = load_formula(ref_vae.ResBlock(32))
    x = g("rb.x", (2, 32, 9, 7))
    save("res_block", x=x, y=rb(x))
    dec = load_formula(ref_vae.Decoder(channels=[64, 32, 32], stages=[1, 2, 1]))
    z = g("dec.z", (2, 8, 6, 5))
    save("decoder_tiny", z=z, y=dec(z))
    full = load_formula(ref_vae.Decoder())
    z = g("decfull.z", (1, 8, 32, 32))
    y = full(z)
    img = torch.clamp(y, -1, 1)
    u8 = (img[0].cpu().numpy() * 127.5 + 127.5).astype(np.uint8).transpose(1, 2, 0)
    save("decoder_full", z=z, y_sub=y[:, :, ::4, ::4], y_rows=y[:, :, 100:104, :],
         y_norm=y.double().norm(), y_mean=y.double().mean(dim=(0, 2, 3)), u8_rows=u8[100:104])


@torch.no_grad()
def encoder_cases():
    enc = load_formula(ref_vae.Encoder(channels=[32, 64, 32], stages=[1, 2, 1]))
    x = g("enc.x", (2, 3, 16, 24))
    save("encoder_tiny", x=x, z=enc(x))
    full = load_formula(ref_vae.Encoder())
    x = g("encfull.x", (1, 3, 128, 128))
    save("encoder_full", x=x, z=full(x))


def decoder_bwd_cases():
    """Decoder backward (vae.py:99-132) through the REFERENCE's autograd: a three-stage decoder with 64-channel first stage (the wide
    weight-gradient route needs multiples of 128 -- the tiny net takes the narrow one; a 128-channel two-stage net takes the wide one),
    loss = sum(y * g) with a fixed g: every parameter gradient and dL/dz."""
    arrs = {}
    for tag, kw, zshape in (("a", dict(channels=[64, 32, 32], stages=[1, 2, 1]), (2, 8, 8, 4)),
                            ("b", dict(channels=[128, 128], stages=[1, 1]), (1, 8, 8, 8))):
        dec = load_formula(ref_vae.Decoder(**kw))
        z = g("decb.z" + tag, zshape).clone().requires_grad_()
        y = dec(z)
        gy = g("decb.g" + tag, tuple(y.shape))
        (y * gy).sum().backward()
        arrs["z_" + tag], arrs["g_" + tag], arrs["y_" + tag], arrs["dz_" + tag] = z.detach(), gy, y.detach(), z.grad
        names = []
        for k, p_ in dec.named_parameters():
            if p_.grad is not None:
                names.append(k)
                if tag == "a":
                    arrs["grad_%s_%s" % (tag, k)] = p_.grad
                else:                                    # wide net: norm + a corner of every gradient (keeps the fixture small)
                    arrs["gradnorm_%s_%s" % (tag, k)] = p_.grad.double().norm()
                    arrs["gradslice_%s_%s" % (tag, k)] = p_.grad.reshape(p_.grad.shape[0], -1)[:32, :96].clone()
        arrs["names_" + tag] = np.array(names)
    save("decoder_bwd", **arrs)


def decoder_oc_cases():
    """Decoder(output_channels=4) and (=1) (vae.py:100-114; every script uses 3): output, dL/dz and every parameter-gradient norm of
    loss = sum(y * g) through the reference's autograd."""
    arrs = {}
    for oc in (4, 1):
        dec = load_formula(ref_vae.Decoder(output_channels=oc, channels=[64, 32], stages=[1, 1]), salt=6)
        z = g("decoc.z%d" % oc, (2, 8, 8, 4)).clone().requires_grad_()
        y = dec(z)
        gy = g("decoc.g%d" % oc, tuple(y.shape))
        (y * gy).sum().backward()
        names = [k for k, p_ in dec.named_parameters() if p_.grad is not None]
        pd = dict(dec.named_parameters())
        arrs.update({"z_%d" % oc: z.detach(), "g_%d" % oc: gy, "y_%d" % oc: y.detach(), "dz_%d" % oc: z.grad, "names_%d" % oc: np.array(names),
                     "gradnorms_%d" % oc: np.asarray([float(pd[k].grad.double().norm()) for k in names]),
                     "grad_to_rgb_%d" % oc: dec.stages[0].to_rgb.weight.grad, "grad_to_rgb_bias_%d" % oc: dec.stages[1].to_rgb.bias.grad})
    save("decoder_oc", **arrs)


def vae_bwd_cases():
    """VAE.calclate_loss (vae.py:36-43) WITH gradients through the reference's autograd: (loss_recon + loss_reg).backward() on a tiny
    encoder / decoder / 512-entry codebook, latent noise replayed; norm + a corner of every parameter gradient."""
    enc = load_formula(ref_vae.Encoder(channels=[32, 64, 32], stages=[1, 2, 1]))
    dec = load_formula(ref_vae.Decoder(channels=[64, 32, 32], stages=[1, 2, 1]))
    vq2 = ref_vae.VectorQuantizer(num_embeddings=512, dim=8)
    with torch.no_grad():
        vq2.embeddings.copy_(g("vq2.emb", (512, 8)) * 0.2)
    vae = ref_vae.VAE(enc, dec, vq2)
    img = g("vaeb.x", (2, 3, 32, 32))
    torch.manual_seed(12)
    st = torch.get_rng_state()
    with torch.no_grad():
        z0 = enc(img)
    noise = torch.randn(z0.shape)
    torch.set_rng_state(st)
    loss_recon, loss_reg, y = vae.calclate_loss(img, noise_gain=0.1)
    (loss_recon + loss_reg).backward()
    arrs = dict(x=img, noise=noise, emb=vq2.embeddings.detach().clone(), loss_recon=loss_recon.detach(), loss_reg=loss_reg.detach(), y=y.detach())
    names = []
    for k, p_ in vae.named_parameters():
        if p_.grad is not None:
            names.append(k)
            arrs["gradnorm_" + k] = p_.grad.double().norm()
            arrs["gradslice_" + k] = p_.grad.reshape(p_.grad.shape[0], -1)[:32, :96].clone()
    arrs["names"] = np.array(names)
    save("vae_bwd", **arrs)


def discriminator_cases():
    """Discriminator (vae.py:134-171, default widths 32 / 48 / 48 / 96) through the reference's autograd: ``calclate_logit`` with
    the generator-side hinge of train_vae.py:113 (relu(-logit)) and ``calclate_logit_and_feature_matching`` (logit + feature
    distance); every parameter gradient (norm + a corner) and the gradient at the fake batch."""
    arrs = {}
    for tag in ("logit", "fm"):
        disc = load_formula(ref_vae.Discriminator(), salt=3)
        fake = g("disc.fake" + tag, (2, 3, 32, 32)).clone().requires_grad_()
        real = g("disc.real" + tag, (2, 3, 32, 32))
        if tag == "logit":
            logit = disc.calclate_logit(fake)
            loss = torch.nn.functional.relu(1 - logit) * 0.5 + logit * 0.25          # both sides of the hinge stay live
            arrs["logit_" + tag] = logit.detach()
        else:
            logit, feat = disc.calclate_logit_and_feature_matching(fake, real.clone())
            loss = logit * 0.5 + feat
            arrs["logit_" + tag], arrs["feat_" + tag] = logit.detach(), feat.detach()
        loss.backward()
        arrs["fake_" + tag], arrs["real_" + tag], arrs["dfake_" + tag] = fake.detach(), real, fake.grad
        names = []
        for k, p_ in disc.named_parameters():
            if p_.grad is not None:
                names.append(k)
                arrs["gradnorm_%s_%s" % (tag, k)] = p_.grad.double().norm()
                arrs["gradslice_%s_%s" % (tag, k)] = p_.grad.reshape(p_.grad.shape[0], -1)[-32:, -96:].clone()   # the LAST channels
        arrs["names_" + tag] = np.array(names)
    save("discriminator", **arrs)


def vq_cases():
    """VectorQuantizer (vae.py:7-26) on its default codebook size (8192 x 8): indices, embedding rows, the two-sided L1 loss and its
    gradients, plus VAE.calclate_loss's forward on a tiny encoder / decoder with the noise replayed.  The codebook gets three
    adversarial rows: two exact duplicates (the FIRST must win) and one row equal to a query (distance 0, the clamp path)."""
    vq = ref_vae.VectorQuantizer()
    emb = g("vq.emb", (8192, 8))
    x = g("vq.x", (2, 1024, 8)) * 1.3
    emb[5000] = emb[77]                                  # exact tie: index 77 wins wherever row 77 is nearest
    x[0, 5] = emb[77] * 1.0001                           # ... and make it nearest for one query
    x[1, 9] = emb[4242]                                  # distance exactly 0
    with torch.no_grad():
        vq.embeddings.copy_(emb)
    xr = x.clone().requires_grad_()
    idx = vq.quantize(xr)
    loss = vq.calculate_loss(xr)
    loss.backward()
    arrs = dict(emb=emb, x=x, idx=idx, e_rows=vq.embed(idx).detach()[:, :16], loss=loss.detach(), dx=xr.grad,
                demb_norm=vq.embeddings.grad.double().norm(), demb_rows=vq.embeddings.grad[[77, 4242, 5000, 0, 8191]],
                demb_nonzero=(vq.embeddings.grad.abs().sum(1) > 0).sum())
    # VAE.calclate_loss forward (vae.py:36-43) on a tiny encoder / decoder; the latent noise is replayed from the torch seed
    enc = load_formula(ref_vae.Encoder(channels=[32, 64, 32], stages=[1, 2, 1]))
    dec = load_formula(ref_vae.Decoder(channels=[64, 32, 32], stages=[1, 2, 1]))
    vq2 = ref_vae.VectorQuantizer(num_embeddings=512, dim=8)
    with torch.no_grad():
        vq2.embeddings.copy_(g("vq2.emb", (512, 8)) * 0.2)
    vae = ref_vae.VAE(enc, dec, vq2)
    img = g("vae.x", (2, 3, 32, 32))
    torch.manual_seed(11)
    st = torch.get_rng_state()
    with torch.no_grad():
        z0 = enc(img)
    noise = torch.randn(z0.shape)
    torch.set_rng_state(st)
    with torch.no_grad():
        loss_recon, loss_reg, y = vae.calclate_loss(img, noise_gain=0.1)
    arrs.update(vae_x=img, vae_noise=noise, vae_emb=vq2.embeddings.detach(), vae_loss_recon=loss_recon, vae_loss_reg=loss_reg, vae_y=y)
    save("vq", **arrs)


def stem_cases():
    """stem_size = 2 (unet.py:75-78: patchify conv in, ConvTranspose2d out), pixel-space tiny net: eval forward, and one
    calculate_loss backward in train mode (all gradient norms, the two stem / head weight gradients and the head bias gradient)."""
    net = load_formula(ref_unet.UNet(input_channels=3, stages=[1, 2], channels=[32, 64], stem_size=2))
    x = g("stem2.x", (3, 3, 32, 24))
    t = torch.tensor([5, 5, 700])
    arrs = dict(x=x, t=t)
    net.eval()
    random.seed(2)
    with torch.no_grad():
        arrs["y_eval"] = net(x, t)
    net.train()
    d = ref_ddpm.DDPM(model=net)
    xl = g("stem2.loss.x", (4, 3, 16, 16))
    torch.manual_seed(5)
    random.seed(5)
    st = torch.get_rng_state()
    tl = torch.randint(low=1, high=1000, size=(4,))
    el = torch.randn(4, 3, 16, 16)
    torch.set_rng_state(st)
    with Trace() as tr:
        loss = d.calculate_loss(xl)
    loss.backward()
    names, norms = [], []
    for k, p in net.named_parameters():
        names.append(k)
        norms.append(-1.0 if p.grad is None else float(p.grad.double().norm()))
    arrs.update(loss_x=xl, loss_t=tl, loss_e=el, loss=loss.detach(), trace=tr.encoded(), grad_names=np.asarray(names), grad_norms=np.asarray(norms),
                grad_encoder_first_weight=net.encoder_first.weight.grad, grad_decoder_last_weight=net.decoder_last.weight.grad,
                grad_decoder_last_bias=net.decoder_last.bias.grad)
    # Discriminator(stem_size=2) (vae.py:135-137): logit + feature distance, gradient at the fake batch, every parameter-gradient norm
    disc = load_formula(ref_vae.Discriminator(channels=[32, 64], stages=[1, 1], stem_size=2), salt=4)
    fake = g("stem2.disc.fake", (2, 3, 32, 32)).clone().requires_grad_()
    real = g("stem2.disc.real", (2, 3, 32, 32))
    logit, feat = disc.calclate_logit_and_feature_matching(fake, real.clone())
    (logit * 0.5 + feat).backward()
    dn = [k for k, p_ in disc.named_parameters() if p_.grad is not None]
    arrs.update(disc_fake=fake.detach(), disc_real=real, disc_logit=logit.detach(), disc_feat=feat.detach(), disc_dfake=fake.grad,
                disc_names=np.asarray(dn), disc_gradnorms=np.asarray([float(dict(disc.named_parameters())[k].grad.double().norm()) for k in dn]),
                disc_grad_input_layer_weight=disc.input_layer.weight.grad)
    save("unet_stem2", **arrs)


if __name__ == "__main__":
    if "--decoder-oc-only" in sys.argv:
        decoder_oc_cases()
        sys.exit(0)
    if "--stem-only" in sys.argv:
        stem_cases()
        sys.exit(0)
    if "--vq-only" in sys.argv:
        vq_cases()
        sys.exit(0)
    if "--loss-full-only" in sys.argv:
        loss_full_cases()
        sys.exit(0)
    if "--encoder-only" in sys.argv:
        encoder_cases()
        sys.exit(0)
    if "--discriminator-only" in sys.argv:
        discriminator_cases()
        sys.exit(0)
    if "--decoder-bwd-only" in sys.argv:
        decoder_bwd_cases()
        vae_bwd_cases()
        sys.exit(0)
    tables()
    schedule()
    module_cases()
    tiny, full = unet_cases()
    sample_cases(tiny, full)
    loss_cases(tiny)
    stem_cases()
    decoder_oc_cases()
    loss_full_cases()
    vae_cases()
    encoder_cases()
    decoder_bwd_cases()
    vae_bwd_cases()
    discriminator_cases()
    vq_cases()
