"""The CPU oracle against the golden vectors generated from the imported reference
(tests/golden/make_golden.py).  CPU only; this is what pins the oracle."""
import random

import numpy as np
import pytest
import torch

from conftest import T, decode_trace, load_golden, max_rel, rel_l2
from oracle import ldm_oracle as O

TOL = 2e-6          # fp32 self-noise of the reference against itself is ~5e-7

TINY = dict(stages=(1, 2, 3, 2), channels=(32, 64, 96, 128))


def test_tables_bit_exact():
    g = load_golden("tables")
    for c, h, w in [(32, 7, 5), (128, 32, 32), (1024, 4, 4)]:
        pe = O.positional_table(c, h, w)[0]
        assert torch.equal(pe, T(g["pe_%d_%d_%d" % (c, h, w)]))
    steps = T(g["te_steps"]).long()
    for c in (32, 128, 1024):
        assert torch.equal(O.time_table(c, steps), T(g["te_%d" % c]))


def test_schedule_bit_exact():
    g = load_golden("schedule")
    beta, alpha_bar, alpha_cum = O.schedule_tables()
    assert torch.equal(beta, T(g["beta"]))
    assert torch.equal(alpha_bar, T(g["alpha_bar"]))
    assert torch.equal(alpha_cum, T(g["alpha_cum"]))
    for n in (5, 20, 50):
        steps, nxt = O.ddim_steps(n)
        assert steps == [int(v) for v in g["steps_%d" % n]]
        assert nxt == [0] + steps[:-1]
    assert O.ddim_steps(50)[0][:4] == [0, 20, 40, 61] and O.ddim_steps(50)[0][-2:] == [978, 999]


def test_channel_norm():
    g = load_golden("channel_norm")
    assert rel_l2(O.channel_norm(T(g["x"])), T(g["y"])) < TOL


def test_encodings():
    g = load_golden("encodings")
    sd = O.formula_state({"proj1.weight": (128, 64, 1, 1), "proj1.bias": (128,),
                          "proj2.weight": (64, 128, 1, 1), "proj2.bias": (64,)})
    x = T(g["x"])
    mul, bias = O.encodings_film(sd, "", 32, x.shape[2], x.shape[3], T(g["t"]))
    assert rel_l2(x * mul + bias, T(g["y"])) < TOL


def test_random_moe_and_rng_equivalence():
    g = load_golden("random_moe")
    shapes = {}
    O.reglu_shapes(shapes, "general.", 32)
    for e in range(4):
        O.reglu_shapes(shapes, "experts.%d." % e, 32)
    sd = O.formula_state(shapes)
    x = T(g["x"])
    assert rel_l2(O.reglu(sd, "general.", x), T(g["general"])) < TOL
    for seed in (0, 1, 7):
        random.seed(seed)
        picks = random.sample(range(4), 2)
        assert picks == [int(v) for v in g["picks_%d" % seed]]
        random.seed(seed)
        assert rel_l2(O.random_moe(sd, "", x), T(g["y_%d" % seed])) < TOL


def test_gconv():
    g = load_golden("gconv")
    sd = O.formula_state({"weight": (64, 32, 3, 3), "bias": (64,)})
    y = torch.nn.functional.conv2d(T(g["x"]), sd["weight"], sd["bias"], padding=1, groups=2)
    assert rel_l2(y, T(g["y"])) < TOL


@pytest.mark.parametrize("shift", [0, 3])
@pytest.mark.parametrize("hw", [(8, 8), (16, 16), (12, 12), (7, 9), (4, 4), (6, 6), (32, 32)])
def test_window_attention(shift, hw):
    g = load_golden("window_attention")
    shapes = {}
    O.mha_shapes(shapes, "attention.", 64)
    sd = O.formula_state(shapes, gain=2.0)
    x = T(g["x_%d_%d" % hw])
    y = O.window_attention(sd, "", x, 6, shift)
    assert rel_l2(y, T(g["y_s%d_%d_%d" % (shift, hw[0], hw[1])])) < TOL


@pytest.mark.parametrize("attn,shift", [(1, 3), (1, 0), (0, 0)])
def test_swin_block(attn, shift):
    g = load_golden("swin_block_a%d_s%d" % (attn, shift))
    shapes = {}
    O.swin_block_shapes(shapes, "", 64, bool(attn))
    sd = O.formula_state(shapes)
    x, t = T(g["x"]), T(g["t"])
    random.seed(5)
    dec = []
    y = O.swin_block(sd, "", x, t, shift, bool(attn), False, dec)
    assert rel_l2(y, T(g["y_eval"])) < TOL
    assert [("s", d[1:]) for d in dec] == decode_trace(g["trace_eval"])
    for seed in (0, 3):
        random.seed(seed)
        y = O.swin_block(sd, "", x, t, shift, bool(attn), True)
        assert rel_l2(y, T(g["y_train_%d" % seed])) < TOL


def _trace_of(decisions, training):
    out = []
    for d in decisions:
        if d[0] == "skip":
            out.append("skip")
        else:
            out.append(d[1:])
    return out


def _golden_trace(arr, training):
    ev = decode_trace(arr)
    out, i = [], 0
    while i < len(ev):
        if ev[i][0] == "r":
            if ev[i][1] <= 0.25:
                out.append("skip")
                i += 1
                continue
            i += 1
        out.append(ev[i][1])
        i += 1
    return out


def test_unet_tiny():
    g = load_golden("unet_tiny")
    sd = O.formula_state(O.unet_state_shapes(8, **TINY))
    x, t = T(g["x"]), T(g["t"])
    random.seed(11)
    dec = []
    y = O.unet_forward(sd, x, t, training=False, decisions=dec, **TINY)
    assert rel_l2(y, T(g["y_eval"])) < TOL
    assert _trace_of(dec, False) == _golden_trace(g["trace_eval"], False)
    for seed in (0, 1):
        random.seed(seed)
        dec = []
        y = O.unet_forward(sd, x, t, training=True, decisions=dec, **TINY)
        assert rel_l2(y, T(g["y_train_%d" % seed])) < TOL
        assert _trace_of(dec, True) == _golden_trace(g["trace_train_%d" % seed], True)


def test_unet_tiny_pixel_space():
    g = load_golden("unet_tiny3")
    cfg = dict(stages=(1, 2), channels=(32, 64))
    sd = O.formula_state(O.unet_state_shapes(3, **cfg))
    random.seed(2)
    y = O.unet_forward(sd, T(g["x"]), T(g["t"]), training=False, **cfg)
    assert rel_l2(y, T(g["y_eval"])) < TOL


def test_unet_stem_size_2():
    """stem_size = 2 (unet.py:75-78): stride-2 patchify conv in, ConvTranspose2d out."""
    g = load_golden("unet_stem2")
    cfg = dict(stages=(1, 2), channels=(32, 64))
    sd = O.formula_state(O.unet_state_shapes(3, stem_size=2, **cfg))
    assert sd["encoder_first.weight"].shape == (32, 3, 2, 2) and sd["decoder_last.weight"].shape == (32, 3, 2, 2)
    random.seed(2)
    y = O.unet_forward(sd, T(g["x"]), T(g["t"]), training=False, **cfg)
    assert y.shape == tuple(g["y_eval"].shape) and rel_l2(y, T(g["y_eval"])) < TOL


def test_unet_full_size():
    g = load_golden("unet_full")
    shapes = O.unet_state_shapes()
    assert len(shapes) == 1376 and sum(int(np.prod(s)) for s in shapes.values()) == 385718536
    sd = O.formula_state(shapes)
    x, t = T(g["x"]), T(g["t"])
    with torch.no_grad():
        random.seed(0)
        y = O.unet_forward(sd, x, t, training=False)
        assert rel_l2(y, T(g["y_eval"])) < TOL
        random.seed(0)
        y = O.unet_forward(sd, x, t, training=True)
        assert rel_l2(y, T(g["y_train_0"])) < TOL


def test_ddim_sample_tiny():
    g = load_golden("sample_tiny")
    sd = O.formula_state(O.unet_state_shapes(8, **TINY))
    for mode in ("train", "eval"):
        for steps in (5, 50):
            x0 = O.ddim_sample(sd, (2, 8, 32, 32), seed=0, num_steps=steps, training=(mode == "train"),
                               unet_kwargs=TINY, prefix="")
            ref = T(g["x0_%s_%d" % (mode, steps)])
            assert rel_l2(x0, ref) < 2e-5 and max_rel(x0, ref) < 1e-4, (mode, steps)
    torch.manual_seed(0)
    assert torch.equal(torch.randn(2, 8, 32, 32), T(g["xT"]))


def test_ddpm_loss_tiny():
    g = load_golden("loss_tiny")
    sd = O.formula_state(O.unet_state_shapes(8, **TINY))
    random.seed(3)
    with torch.no_grad():
        loss = O.ddpm_loss(sd, T(g["x"]), t=T(g["t"]), e=T(g["e"]), training=True, unet_kwargs=TINY, prefix="")
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))


def test_res_block_and_decoder():
    g = load_golden("res_block")
    sd = O.formula_state({"c1.weight": (32, 32, 3, 3), "c1.bias": (32,), "c2.weight": (32, 32, 3, 3), "c2.bias": (32,)})
    assert rel_l2(O.res_block(sd, "", T(g["x"])), T(g["y"])) < TOL
    g = load_golden("decoder_tiny")
    cfg = dict(channels=(64, 32, 32), stages=(1, 2, 1))
    sd = O.formula_state(O.decoder_state_shapes(**cfg))
    y = O.vae_decode(sd, T(g["z"]), stages=cfg["stages"])
    assert rel_l2(y, T(g["y"])) < TOL


@pytest.mark.parametrize("oc", [4, 1])
def test_decoder_output_channels(oc):
    """Decoder(output_channels=oc) (vae.py:100-114)."""
    g = load_golden("decoder_oc")
    sd = O.formula_state(O.decoder_state_shapes(output_channels=oc, channels=(64, 32), stages=(1, 1)), salt=6)
    with torch.no_grad():
        y = O.vae_decode(sd, T(g["z_%d" % oc]), stages=(1, 1))
    assert y.shape == tuple(g["y_%d" % oc].shape) and rel_l2(y, T(g["y_%d" % oc])) < TOL


def test_decoder_full_size():
    g = load_golden("decoder_full")
    shapes = O.decoder_state_shapes()
    assert len(shapes) == 50 and sum(int(np.prod(s)) for s in shapes.values()) == 13233871
    sd = O.formula_state(shapes)
    with torch.no_grad():
        y = O.vae_decode(sd, T(g["z"]))
    assert rel_l2(y[:, :, ::4, ::4], T(g["y_sub"])) < TOL
    assert rel_l2(y[:, :, 100:104, :], T(g["y_rows"])) < TOL
    assert abs(float(y.double().norm()) - float(g["y_norm"])) < 1e-5 * float(g["y_norm"])
    u8 = O.to_uint8_hwc(y)[0]
    diff = np.abs(u8[100:104].astype(int) - g["u8_rows"].astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3       # truncation can flip at exact boundaries


def test_ddim_sample_full_size():
    g = load_golden("sample_full")
    sd = O.formula_state(O.unet_state_shapes())
    x0 = O.ddim_sample(sd, (1, 8, 32, 32), seed=0, num_steps=3, training=True, prefix="")
    assert rel_l2(x0, T(g["x0_train_3"])) < 2e-5
    x0 = O.ddim_sample(sd, (1, 8, 32, 32), seed=0, num_steps=3, training=False, prefix="")
    assert rel_l2(x0, T(g["x0_eval_3"])) < 2e-5
    x0 = O.ddim_sample(sd, (1, 8, 32, 32), seed=0, num_steps=50, training=True, prefix="")
    ref = T(g["x0_train_50"])
    assert rel_l2(x0, ref) < 1e-4 and max_rel(x0, ref) < 1e-3


def test_ddpm_loss_gradients_tiny():
    """Autograd through the oracle reproduces the reference's per-parameter gradient norms (incl. which
    parameters get no gradient at all, and the detached float mask of the shifted windows)."""
    g = load_golden("loss_tiny")
    sd = {k: v.requires_grad_() for k, v in O.formula_state(O.unet_state_shapes(8, **TINY)).items()}
    random.seed(3)
    loss = O.ddpm_loss(sd, T(g["x"]), t=T(g["t"]), e=T(g["e"]), training=True, unet_kwargs=TINY, prefix="")
    loss.backward()
    norms = dict(zip([str(n) for n in g["grad_names"]], g["grad_norms"]))
    for k, v in sd.items():
        ref = norms[k]
        if ref < 0:
            assert v.grad is None, k
        else:
            assert abs(float(v.grad.double().norm()) - ref) < 1e-5 * max(ref, 1e-9) + 1e-12, k
    assert rel_l2(sd["encoder_first.weight"].grad, T(g["grad_encoder_first_weight"])) < 1e-5


def test_encoder_tiny_and_full():
    g = load_golden("encoder_tiny")
    cfg = dict(channels=(32, 64, 32), stages=(1, 2, 1))
    sd = O.formula_state(O.encoder_state_shapes(**cfg))
    assert rel_l2(O.vae_encode(sd, T(g["x"]), stages=cfg["stages"]), T(g["z"])) < TOL
    g = load_golden("encoder_full")
    shapes = O.encoder_state_shapes()
    assert sum(int(np.prod(s)) for s in shapes.values()) == 12714888
    with torch.no_grad():
        assert rel_l2(O.vae_encode(O.formula_state(shapes), T(g["x"])), T(g["z"])) < TOL


def test_discriminator_logit_feature_matching_and_gradients():
    """vae.py:134-171 through the oracle's restatement + torch autograd vs the reference's own autograd (golden ``discriminator``)."""
    g = load_golden("discriminator")
    shapes = O.discriminator_state_shapes()
    assert sum(int(np.prod(s)) for s in shapes.values()) == 569764
    for tag in ("logit", "fm"):
        sd = {k: v.clone().requires_grad_() for k, v in O.formula_state(shapes, salt=3).items()}
        fake = T(g["fake_" + tag]).clone().requires_grad_()
        if tag == "logit":
            logit = O.discriminator_logit(sd, fake)
            loss = torch.relu(1 - logit) * 0.5 + logit * 0.25
        else:
            logit, feat = O.discriminator_logit_and_feature_matching(sd, fake, T(g["real_" + tag]))
            assert abs(float(feat) - float(g["feat_" + tag])) < TOL * abs(float(g["feat_" + tag]))
            loss = logit * 0.5 + feat
        assert abs(float(logit) - float(g["logit_" + tag])) < 1e-6
        loss.backward()
        assert rel_l2(fake.grad, T(g["dfake_" + tag])) < 1e-5
        for k in g["names_" + tag]:
            k = str(k)
            ref = float(g["gradnorm_%s_%s" % (tag, k)])
            assert abs(float(sd[k].grad.double().norm()) - ref) < 1e-5 * ref + 1e-12, k
            sl = sd[k].grad.reshape(sd[k].shape[0], -1)[-32:, -96:]
            assert rel_l2(sl, T(g["gradslice_%s_%s" % (tag, k)])) < 1e-5, k
