"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A from-the-math restatement (torch-CPU, fp32) of the latent-diffusion hot path of
uthree/ldm-image-generator: UNet forward, DDIM sampling loop, DDPM training loss
and the VAE decoder.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this file, and only as the
checker / reported CPU baseline.  The product (``ldm_image_generator_amd``)
never imports it and has no CPU fallback.

The reference's arithmetic lives in third-party PyTorch (unpinned by the
reference; torch 2.10.0+rocm7.0 in this image).  The reference ships no tests
and no golden vectors, so this oracle is PINNED BY GENERATED FIXTURES: the
reference was imported in the build container (``tests/golden/make_golden.py``,
committed) and its outputs on formula weights are stored under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function below
against them.

Style: pure functions over a flat ``state_dict`` (the reference's checkpoint
ABI) plus a key prefix -- no nn.Module mirrors.  All activations
are NCHW like the reference.  Each function cites the reference lines it
restates (paths relative to /root/reference).
"""
import math
import random

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# sinusoidal.py
# --------------------------------------------------------------------------
def positional_table(channels, height, width, dtype=torch.float32):
    """sinusoidal.py:12-19 (PositionalEncoding2d) -> [1, C, H, W].

    Channels [0, C/4): sin((i/H)*pi*f_k); [C/4, C/2): cos(same); the upper half
    repeats the construction along the width.  f_k = 1 / 2**(k/(C/4)).
    """
    q = channels // 4
    rows = torch.arange(height, dtype=dtype).reshape(1, 1, height, 1) / height
    cols = torch.arange(width, dtype=dtype).reshape(1, 1, 1, width) / width
    freq = 1 / (2 ** (torch.arange(q).reshape(1, q, 1, 1) / q))
    rv = rows * math.pi * freq
    cv = cols * math.pi * freq
    ev = torch.cat([torch.sin(rv), torch.cos(rv)], dim=1).expand(1, 2 * q, height, width)
    eh = torch.cat([torch.sin(cv), torch.cos(cv)], dim=1).expand(1, 2 * q, height, width)
    return torch.cat([ev, eh], dim=1)


def time_table(channels, t, max_timesteps=10000):
    """sinusoidal.py:31-38 (TimeEncoding2d) -> [B, C] (constant over H, W).

    t is an integer tensor; (t * pi) is formed in float32 first, then multiplied
    by f_k = 1 / 10000**(k/(C/2)); sin fills [0, C/2), cos fills [C/2, C).
    """
    half = channels // 2
    tt = t.reshape(-1, 1).expand(t.shape[0], half)
    freq = (1 / (max_timesteps ** (torch.arange(half) / half))).reshape(1, half)
    arg = tt * math.pi * freq
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)


# --------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------
def _pointwise(x, w, b):
    """1x1 convolution (nn.Conv2d(cin, cout, 1, 1, 0)) as a channel matmul."""
    cout = w.shape[0]
    y = torch.einsum("oc,nchw->nohw", w.reshape(cout, -1), x)
    return y + b.reshape(1, cout, 1, 1)


def channel_norm(x, eps=1e-4):
    """modules.py:23-25: per-pixel over C, UNBIASED variance, eps inside the sqrt."""
    mu = x.mean(dim=1, keepdim=True)
    var = x.var(dim=1, keepdim=True)          # unbiased (divides by C-1)
    return (x - mu) / torch.sqrt(var + eps)


def encodings_film(sd, p, channels, height, width, t):
    """unet.py:18-21: (mul, bias) of the FiLM, each [B, C, H, W] (B = len(t)).

    The MLP input is cat[positional, time] (2C channels); the position half is
    sample-independent, so samples with equal t share one evaluation (pure
    de-duplication -- per-sample results are those of the reference).
    """
    uniq, inv = torch.unique(t, return_inverse=True)
    pe = positional_table(channels, height, width).expand(uniq.shape[0], channels, height, width)
    te = time_table(channels, uniq).reshape(uniq.shape[0], channels, 1, 1).expand(-1, -1, height, width)
    e = torch.cat([pe, te], dim=1)
    hdn = torch.relu(_pointwise(e, sd[p + "proj1.weight"], sd[p + "proj1.bias"]))
    out = _pointwise(hdn, sd[p + "proj2.weight"], sd[p + "proj2.bias"])
    out = out[inv]
    return out[:, :channels], out[:, channels:]


def reglu(sd, p, x):
    """modules.py:14-15: c(a(x) * relu(b(x))), all 1x1."""
    a = _pointwise(x, sd[p + "a.weight"], sd[p + "a.bias"])
    b = _pointwise(x, sd[p + "b.weight"], sd[p + "b.bias"])
    return _pointwise(a * torch.relu(b), sd[p + "c.weight"], sd[p + "c.bias"])


def random_moe(sd, p, x, picks=None, num_experts=4):
    """modules.py:34-36: general + two experts drawn with Python's global RNG.

    ``random.sample(list(experts), 2)`` consumes the generator exactly like
    ``random.sample(range(4), 2)`` (CPython's pool algorithm depends only on
    len(population) and k).
    """
    if picks is None:
        picks = random.sample(range(num_experts), 2)
    y = reglu(sd, p + "general.", x)
    for e in picks:
        y = y + reglu(sd, p + "experts.%d." % e, x)
    return y


# --------------------------------------------------------------------------
# attention.py
# --------------------------------------------------------------------------
def multihead_self_attention(sd, p, tokens, key_bias=None, key_mask=None, head_dim=32):
    """nn.MultiheadAttention(C, C/32, batch_first=True)(x, x, x, key_padding_mask=...)
    as called at attention.py:82 (published algorithm: torch nn/functional.py
    multi_head_attention_forward).  tokens [Nb, L, C].

    key_mask  [Nb, L] bool : True keys get -inf before the softmax;
    key_bias  [Nb, L] float: added to every logit of that key (what a FLOAT
                             key_padding_mask means to torch -- attention.py:40's
                             typo turns the shifted-window mask into this).
    q is scaled by sqrt(1/head_dim) before q.k^T.
    """
    nb, length, c = tokens.shape
    heads = c // head_dim
    w_in, b_in = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    qkv = tokens @ w_in.t() + b_in
    q, k, v = qkv.split(c, dim=2)

    def split_heads(z):
        return z.reshape(nb, length, heads, head_dim).permute(0, 2, 1, 3)

    q, k, v = split_heads(q), split_heads(k), split_heads(v)
    logits = (q * math.sqrt(1.0 / head_dim)) @ k.transpose(2, 3)        # [Nb, h, L, L]
    if key_mask is not None:
        logits = logits.masked_fill(key_mask.reshape(nb, 1, 1, length), float("-inf"))
    if key_bias is not None:
        logits = logits + key_bias.reshape(nb, 1, 1, length)
    att = torch.softmax(logits, dim=-1) @ v                              # [Nb, h, L, d]
    att = att.permute(0, 2, 1, 3).reshape(nb, length, c)
    return att @ sd[p + "out_proj.weight"].t() + sd[p + "out_proj.bias"]


def _to_windows(x, ws):
    """attention.py:62-66 as one reshape: [N,C,Hp,Wp] -> [nwin*N, C, ws, ws] with
    window index (col_window * nwin_h + row_window) major, sample minor."""
    n, c, hp, wp = x.shape
    nh, nw = hp // ws, wp // ws
    x = x.reshape(n, c, nh, ws, nw, ws).permute(4, 2, 0, 1, 3, 5)
    return x.reshape(nw * nh * n, c, ws, ws)


def _from_windows(xw, n, nh, nw, ws):
    """attention.py:68-71, inverse of _to_windows."""
    c = xw.shape[1]
    x = xw.reshape(nw, nh, n, c, ws, ws).permute(2, 3, 1, 4, 0, 5)
    return x.reshape(n, c, nh * ws, nw * ws)


def window_attention(sd, p, x, window_size=6, shift=0):
    """attention.py:13-60."""
    n, c, h, w = x.shape
    ws = window_size
    if h <= ws and w <= ws:                               # attention.py:15-16, global, no mask
        tok = x.reshape(n, c, h * w).transpose(1, 2)
        y = multihead_self_attention(sd, p + "attention.", tok)
        return y.transpose(1, 2).reshape(n, c, h, w)
    pad_h = (ws - h % ws) % ws                            # attention.py:21-25
    pad_w = (ws - w % ws) % ws
    xp = F.pad(x, (0, pad_w, 0, pad_h))                   # zeros at bottom/right, :27-28
    hp, wp = h + pad_h, w + pad_w
    key_mask = key_bias = None
    if shift != 0:
        xr = torch.roll(xp, (shift, shift), (2, 3))       # :39
        # :40 assigns roll(x) to ``mask``: a FLOAT tensor, rolled twice in total.
        bias_map = torch.roll(xr, (shift, shift), (2, 3))[:, 0:1]       # channel 0 is what :81 keeps
        # attention.py:76-81 slices the mask under torch.no_grad(): no gradient flows through it
        key_bias = _to_windows(bias_map, ws).reshape(-1, ws * ws).detach()
        xp = xr
    else:
        m = torch.zeros(1, 1, hp, wp, dtype=torch.bool)   # :31-35
        m[:, :, h:, :] = True
        m[:, :, :, w:] = True
        key_mask = _to_windows(m.expand(n, 1, hp, wp), ws).reshape(-1, ws * ws)
    xw = _to_windows(xp, ws)                              # [nwin*N, C, ws, ws]
    tok = xw.reshape(xw.shape[0], c, ws * ws).transpose(1, 2)
    y = multihead_self_attention(sd, p + "attention.", tok, key_bias=key_bias, key_mask=key_mask)
    y = y.transpose(1, 2).reshape(xw.shape[0], c, ws, ws)
    y = _from_windows(y, n, hp // ws, wp // ws, ws)
    if shift != 0:
        y = torch.roll(y, (-shift, -shift), (2, 3))       # :55-56
    return y[:, :, :h, :w]                                # :59


# --------------------------------------------------------------------------
# unet.py
# --------------------------------------------------------------------------
def swin_block(sd, p, x, t, shift, attention, training, decisions=None, head_dim=32, window_size=6):
    """unet.py:38-48.  ``decisions`` (optional) is a list that receives
    ("skip",) / ("run", e1, e2) so tests can compare RNG traces."""
    if training and random.random() <= 0.25:              # :39-40 (draw only in training mode)
        if decisions is not None:
            decisions.append(("skip",))
        return x
    n, c, h, w = x.shape
    res = x
    y = channel_norm(x)
    mul, bias = encodings_film(sd, p + "encodings.", c, h, w, t)
    y = y * mul + bias                                    # :22
    picks = random.sample(range(4), 2)                    # modules.py:35
    if decisions is not None:
        decisions.append(("run", picks[0], picks[1]))
    out = random_moe(sd, p + "ffn.", y, picks=picks)
    out = out + F.conv2d(y, sd[p + "conv.weight"], sd[p + "conv.bias"], padding=1, groups=c // head_dim)
    if attention:
        out = out + window_attention(sd, p + "self_attention.", y, window_size, shift)
    return out + res                                      # :47 (cross-attention is dead code, :45)


def swin_stack(sd, p, x, t, num_blocks, attention, training, decisions=None):
    """unet.py:50-66: shift = 3 on even block index; attention only in the last two blocks."""
    for i in range(num_blocks):
        shift = 3 if i % 2 == 0 else 0
        flag = attention and i >= num_blocks - 2
        x = swin_block(sd, "%sblocks.%d." % (p, i), x, t, shift, flag, training, decisions)
    return x


def unet_forward(sd, x, t, stages=(3, 3, 9, 3), channels=(128, 256, 512, 1024), training=True,
                 prefix="", decisions=None):
    """unet.py:89-103.  ``training`` defaults to True because no reference script
    ever calls .eval() -- sampling runs with stochastic depth live."""
    p = prefix
    ns = len(stages)
    w0 = sd[p + "encoder_first.weight"]                   # Conv2d(Cin, C0, stem_size, stem_size, 0)  (unet.py:77)
    stem = w0.shape[-1]
    x = _pointwise(x, w0, sd[p + "encoder_first.bias"]) if stem == 1 else F.conv2d(x, w0, sd[p + "encoder_first.bias"], stride=stem)
    skips = []
    for i in range(ns):
        x = swin_stack(sd, "%sencoder_stages.%d.stage." % (p, i), x, t, stages[i], False, training, decisions)
        if i == ns - 1:
            skips.insert(0, None)
        else:
            skips.insert(0, x)
            q = "%sencoder_stages.%d.ch_conv.0." % (p, i)
            x = F.avg_pool2d(_pointwise(x, sd[q + "weight"], sd[q + "bias"]), 2)      # unet.py:83
    for j in range(ns):
        i = ns - 1 - j                                    # decoder_stages.0 is the deepest (unet.py:87)
        if j > 0:
            q = "%sdecoder_stages.%d.ch_conv.1." % (p, j)
            x = F.interpolate(x, scale_factor=2, mode="nearest")                      # unet.py:85
            x = _pointwise(x, sd[q + "weight"], sd[q + "bias"])
        if skips[j] is not None:
            x = x + skips[j]
        x = swin_stack(sd, "%sdecoder_stages.%d.stage." % (p, j), x, t, stages[i], True, training, decisions)
    # decoder_last = ConvTranspose2d(C0, Cin, 1, 1): weight [C0, Cin, 1, 1]  (unet.py:78)
    w = sd[p + "decoder_last.weight"]
    if stem > 1:                                           # ConvTranspose2d(C0, Cin, stem_size, stem_size, 0): weight [C0, Cin, s, s]
        return F.conv_transpose2d(x, w, sd[p + "decoder_last.bias"], stride=stem)
    y = torch.einsum("co,nchw->nohw", w.reshape(w.shape[0], w.shape[1]), x)
    return y + sd[p + "decoder_last.bias"].reshape(1, -1, 1, 1)


# --------------------------------------------------------------------------
# ddpm.py
# --------------------------------------------------------------------------
def schedule_tables(beta_min=1e-4, beta_max=0.02, num_timesteps=1000):
    """ddpm.py:19-31,73: beta, alpha_bar (prefix products, used by calculate_loss)
    and the cumprod table that ``sample`` uses."""
    beta = torch.linspace(beta_min, beta_max, num_timesteps)
    alpha = 1 - beta
    alpha_bar = torch.Tensor([torch.prod(alpha[:k]) for k in range(1, num_timesteps + 1)])
    alpha_cum = torch.cumprod(1 - beta, dim=0)
    return beta, alpha_bar, alpha_cum


def ddim_steps(num_steps, num_timesteps=1000):
    """ddpm.py:67,72: linspace(0, T-1, n).int(); next = [0] + steps[:-1]."""
    steps = [int(s) for s in torch.linspace(0, num_timesteps - 1, num_steps).int()]
    return steps, [0] + steps[:-1]


def ddim_coefficients(alpha_cum, t, t_next, eta=0.0):
    """ddpm.py:81-85 scalars as fp32 0-dim tensors: sigma, sqrt(1-a_t), sqrt(a_t),
    sqrt(a_next), sqrt(1-a_next-sigma^2)."""
    a_t, a_n = alpha_cum[t], alpha_cum[t_next]
    sigma = eta * torch.sqrt((1 - a_n) / (1 - a_t)) * torch.sqrt(1 - a_t / a_n)
    return sigma, torch.sqrt(1 - a_t), torch.sqrt(a_t), torch.sqrt(a_n), torch.sqrt(1 - a_n - sigma ** 2)


def ddim_sample(sd, x_shape, seed=None, num_steps=20, eta=0.0, training=True, x_init=None,
                unet_kwargs=None, prefix="model.", decisions=None, schedule="linear", noises=None):
    """ddpm.py:51-93 on CPU (use_autocast is a no-op on CPU, ddpm.py:75)."""
    unet_kwargs = unet_kwargs or {}
    if seed is not None:                                  # :56-61
        random.seed(seed)
        torch.manual_seed(seed)
    x = torch.randn(*x_shape) if x_init is None else x_init.clone()
    _, _, alpha_cum = schedule_tables()
    if schedule == "linear":
        steps, steps_next = ddim_steps(num_steps)
    else:                                                 # ddpm.py:68-69: an explicit list of timesteps
        steps = list(schedule)
        steps_next = [0] + steps[:-1]
    with torch.no_grad():
        for t, t_next in zip(reversed(steps), reversed(steps_next)):
            tt = torch.full((x_shape[0],), t)
            e_theta = unet_forward(sd, x, tt, training=training, prefix=prefix, decisions=decisions, **unet_kwargs)
            e = torch.randn(*x_shape)                     # :80 (consumed even when sigma == 0)
            if noises is not None:                        # tests inject the per-step noise (device RNG streams differ)
                e = noises.pop(0)
            sigma, s1, s2, s3, s4 = ddim_coefficients(alpha_cum, t, t_next, eta)
            x_t0 = (x - s1 * e_theta) / s2
            x = x_t0 if t == 0 else s3 * x_t0 + s4 * e_theta + sigma * e
    return x


def ddpm_loss(sd, x, t=None, e=None, training=True, unet_kwargs=None, prefix="model."):
    """ddpm.py:39-48 (L1 loss).  t and e may be injected for reproducible tests."""
    unet_kwargs = unet_kwargs or {}
    _, alpha_bar, _ = schedule_tables()
    if t is None:
        t = torch.randint(low=1, high=1000, size=(x.shape[0],))
    ab = torch.index_select(alpha_bar, 0, t).reshape(-1, 1, 1, 1)
    if e is None:
        e = torch.randn(*x.shape)
    xt = torch.sqrt(ab) * x + torch.sqrt(1 - ab) * e
    e_theta = unet_forward(sd, xt, t, training=training, prefix=prefix, **unet_kwargs)
    return (e_theta - e).abs().mean()


# --------------------------------------------------------------------------
# vae.py (decoder only)
# --------------------------------------------------------------------------
def res_block(sd, p, x):
    """vae.py:60-66: x + lrelu(c2(lrelu(c1(x)))), slope 0.01."""
    y = F.leaky_relu(F.conv2d(x, sd[p + "c1.weight"], sd[p + "c1.bias"], padding=1))
    y = F.leaky_relu(F.conv2d(y, sd[p + "c2.weight"], sd[p + "c2.bias"], padding=1))
    return y + x


def vae_decode(sd, z, stages=(2, 2, 2, 2), prefix=""):
    """vae.py:122-132: sum of bilinearly upsampled per-stage RGB heads (output_layer is dead)."""
    p = prefix
    x = _pointwise(z, sd[p + "input_layer.weight"], sd[p + "input_layer.bias"])
    rgb_out = None
    for s, nblk in enumerate(stages):
        if s > 0:
            x = F.conv_transpose2d(x, sd["%supsamples.%d.weight" % (p, s)], sd["%supsamples.%d.bias" % (p, s)], stride=2)
        for k in range(nblk):
            x = res_block(sd, "%sstages.%d.layers.%d." % (p, s, k), x)
        rgb = _pointwise(x, sd["%sstages.%d.to_rgb.weight" % (p, s)], sd["%sstages.%d.to_rgb.bias" % (p, s)])
        if rgb_out is None:
            rgb_out = rgb
        else:
            rgb_out = F.interpolate(rgb_out, scale_factor=2, mode="bilinear") + rgb
    return rgb_out


def vae_encode(sd, x, stages=(2, 2, 2, 2), prefix=""):
    """vae.py:91-96 (Encoder.forward): 1x1 in -> per stage [ResBlocks -> AvgPool2d(2) -> 1x1] -> 1x1 out."""
    p = prefix
    y = _pointwise(x, sd[p + "input_layer.weight"], sd[p + "input_layer.bias"])
    for s, nblk in enumerate(stages):
        for k in range(nblk):
            y = res_block(sd, "%sstages.%d.seq.%d." % (p, s, k), y)
        if s < len(stages) - 1:
            q = "%sdownsamples.%d.1." % (p, s)
            y = _pointwise(F.avg_pool2d(y, 2), sd[q + "weight"], sd[q + "bias"])
    return _pointwise(y, sd[p + "output_layer.weight"], sd[p + "output_layer.bias"])


def encoder_state_shapes(input_channels=3, latent_channels=8, channels=(64, 128, 256, 512), stages=(2, 2, 2, 2), prefix=""):
    out = {}
    p = prefix
    _conv_keys(out, p + "input_layer.", channels[0], input_channels)
    _conv_keys(out, p + "output_layer.", latent_channels, channels[-1])
    for s, c in enumerate(channels):
        for k in range(stages[s]):
            _conv_keys(out, "%sstages.%d.seq.%d.c1." % (p, s, k), c, c, 3)
            _conv_keys(out, "%sstages.%d.seq.%d.c2." % (p, s, k), c, c, 3)
    for s in range(len(channels) - 1):
        _conv_keys(out, "%sdownsamples.%d.1." % (p, s), channels[s + 1], channels[s])
    return out


def discriminator_features(sd, x, stages=(2, 2, 2, 2), prefix=""):
    """vae.py:149-171 (both ``calclate_logit*`` walk the same layers): the per-stage feature maps after each ResStack and the
    per-stage early-exit maps c(x) (1 channel); Conv2d(c, c', 2, 2) between stages."""
    p = prefix
    w_in = sd[p + "input_layer.weight"]                   # Conv2d(Cin, C0, stem_size, stem_size, 0)  (vae.py:137)
    y = _pointwise(x, w_in, sd[p + "input_layer.bias"]) if w_in.shape[-1] == 1 else F.conv2d(x, w_in, sd[p + "input_layer.bias"], stride=w_in.shape[-1])
    feats, exits = [], []
    for s, nblk in enumerate(stages):
        for k in range(nblk):
            y = res_block(sd, "%sstages.%d.seq.%d." % (p, s, k), y)
        feats.append(y)
        exits.append(_pointwise(y, sd["%searly_exits.%d.weight" % (p, s)], sd["%searly_exits.%d.bias" % (p, s)]))
        if s < len(stages) - 1:
            y = F.conv2d(y, sd["%sdownsamples.%d.weight" % (p, s)], sd["%sdownsamples.%d.bias" % (p, s)], stride=2)
    return feats, exits


def discriminator_logit(sd, fake, stages=(2, 2, 2, 2), prefix=""):
    """vae.py:163-171: sum over stages of the spatial (and batch) mean of the early exit."""
    return sum(e.mean() for e in discriminator_features(sd, fake, stages, prefix)[1])


def discriminator_logit_and_feature_matching(sd, fake, real, stages=(2, 2, 2, 2), prefix=""):
    """vae.py:149-161: the logit of the fake batch and sum over stages of mean |feature(fake) - feature(real)|."""
    ff, fe = discriminator_features(sd, fake, stages, prefix)
    rf, _ = discriminator_features(sd, real, stages, prefix)
    return sum(e.mean() for e in fe), sum((a - b).abs().mean() for a, b in zip(ff, rf))


def discriminator_state_shapes(input_channels=3, channels=(32, 48, 48, 96), stages=(2, 2, 2, 2), prefix=""):
    """registration order of vae.py:135-147: input_layer, stages, then per stage (downsample, early_exit) -- ``downsamples`` and
    ``early_exits`` are appended alternately but live in separate ModuleLists, so state_dict lists all early_exits first."""
    out = {}
    p = prefix
    _conv_keys(out, p + "input_layer.", channels[0], input_channels)
    for s, c in enumerate(channels):
        for k in range(stages[s]):
            _conv_keys(out, "%sstages.%d.seq.%d.c1." % (p, s, k), c, c, 3)
            _conv_keys(out, "%sstages.%d.seq.%d.c2." % (p, s, k), c, c, 3)
    for s, c in enumerate(channels):
        _conv_keys(out, "%searly_exits.%d." % (p, s), 1, c)
    for s in range(len(channels) - 1):
        _conv_keys(out, "%sdownsamples.%d." % (p, s), channels[s + 1], channels[s], 2)
    return out


def to_uint8_hwc(img):
    """sample_ldm.py:75-77: clamp(-1,1) -> *127.5+127.5 -> uint8 TRUNCATION -> HWC."""
    img = torch.clamp(img, -1, 1)
    return (img.numpy() * 127.5 + 127.5).astype("uint8").transpose(0, 2, 3, 1)


# --------------------------------------------------------------------------
# checkpoint ABI: key -> shape, for building formula weights
# --------------------------------------------------------------------------
def _conv_keys(out, p, cout, cin, k=1):
    out[p + "weight"] = (cout, cin, k, k)
    out[p + "bias"] = (cout,)


def reglu_shapes(out, p, c, ffn_mul=1):
    _conv_keys(out, p + "a.", c * ffn_mul, c)
    _conv_keys(out, p + "b.", c * ffn_mul, c)
    _conv_keys(out, p + "c.", c, c * ffn_mul)


def mha_shapes(out, p, c):
    out[p + "in_proj_weight"] = (3 * c, c)
    out[p + "in_proj_bias"] = (3 * c,)
    out[p + "out_proj.weight"] = (c, c)
    out[p + "out_proj.bias"] = (c,)


def swin_block_shapes(out, p, c, attention, head_dim=32):
    """Key ORDER follows module registration order in unet.py:27-36."""
    reglu_shapes(out, p + "ffn.general.", c)
    for e in range(4):
        reglu_shapes(out, p + "ffn.experts.%d." % e, c)
    out[p + "conv.weight"] = (c, head_dim, 3, 3)
    out[p + "conv.bias"] = (c,)
    if attention:
        mha_shapes(out, p + "self_attention.attention.", c)
        mha_shapes(out, p + "cross_attention.attention.", c)      # dead weights, kept for the ABI
    _conv_keys(out, p + "encodings.proj1.", 4 * c, 2 * c)
    _conv_keys(out, p + "encodings.proj2.", 2 * c, 4 * c)


def unet_state_shapes(input_channels=8, stages=(3, 3, 9, 3), channels=(128, 256, 512, 1024), prefix="", stem_size=1):
    out = {}
    p = prefix
    ns = len(stages)
    _conv_keys(out, p + "encoder_first.", channels[0], input_channels)
    out[p + "encoder_first.weight"] = (channels[0], input_channels, stem_size, stem_size)
    out[p + "decoder_last.weight"] = (channels[0], input_channels, stem_size, stem_size)
    out[p + "decoder_last.bias"] = (input_channels,)
    for i in range(ns):
        for b in range(stages[i]):
            swin_block_shapes(out, "%sencoder_stages.%d.stage.blocks.%d." % (p, i, b), channels[i], False)
        if i < ns - 1:
            _conv_keys(out, "%sencoder_stages.%d.ch_conv.0." % (p, i), channels[i + 1], channels[i])
    for j in range(ns):
        i = ns - 1 - j
        for b in range(stages[i]):
            swin_block_shapes(out, "%sdecoder_stages.%d.stage.blocks.%d." % (p, j, b), channels[i],
                              b >= stages[i] - 2)
        if j > 0:
            _conv_keys(out, "%sdecoder_stages.%d.ch_conv.1." % (p, j), channels[i], channels[i + 1])
    return out


def decoder_state_shapes(output_channels=3, latent_channels=8, channels=(512, 256, 128, 64), stages=(2, 2, 2, 2),
                         prefix=""):
    out = {}
    p = prefix
    _conv_keys(out, p + "input_layer.", channels[0], latent_channels)
    _conv_keys(out, p + "output_layer.", output_channels, channels[-1])
    for s, c in enumerate(channels):
        for k in range(stages[s]):
            _conv_keys(out, "%sstages.%d.layers.%d.c1." % (p, s, k), c, c, 3)
            _conv_keys(out, "%sstages.%d.layers.%d.c2." % (p, s, k), c, c, 3)
        _conv_keys(out, "%sstages.%d.to_rgb." % (p, s), output_channels, c)
    for s in range(1, len(channels)):
        out["%supsamples.%d.weight" % (p, s)] = (channels[s - 1], channels[s], 2, 2)
        out["%supsamples.%d.bias" % (p, s)] = (channels[s],)
    return out


def formula_state(shapes, salt=0, gain=1.0):
    """Formula weights for a key->shape map (same values make_golden.py loaded into the reference)."""
    from ldm_image_generator_amd import synth
    import torch as _t
    return synth.fill_state_dict({k: _t.empty(v) for k, v in shapes.items()}, salt=salt, gain=gain)
