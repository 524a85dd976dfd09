"""GPU: VectorQuantizer kernels (csrc/vq.hip) -- indices bit-exact against the reference's own (golden) and against the CPU
oracle on fresh data incl. ties / NaN / empty-ish shapes; embedding rows exact; loss and gradients against the golden."""
import numpy as np
import pytest
import torch

from conftest import T, load_golden, rel_l2
from oracle import ldm_vq_oracle as V

pytestmark = pytest.mark.gpu


def test_vq_quantize_embed_loss_match_reference(gpu_device):
    from ldm_image_generator_amd.vae import VectorQuantizer
    g = load_golden("vq")
    vq = VectorQuantizer().cuda()
    with torch.no_grad():
        vq.embeddings.copy_(T(g["emb"]))
    x = T(g["x"]).cuda().requires_grad_()
    idx = vq.quantize(x)
    assert idx.shape == (2, 1024) and idx.dtype == torch.int64
    assert torch.equal(idx.cpu(), T(g["idx"]))                                  # every index, bit-exact
    assert torch.equal(vq.embed(idx).cpu()[:, :16], T(g["e_rows"]))
    loss = vq.calculate_loss(x)
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-6 * float(g["loss"])
    assert torch.equal(x.grad.cpu(), T(g["dx"]))                                # +-1/n signs: exact
    demb = vq.embeddings.grad.cpu()
    assert rel_l2(demb[[77, 4242, 5000, 0, 8191]], T(g["demb_rows"])) < 1e-6
    assert abs(float(demb.double().norm()) - float(g["demb_norm"])) < 1e-5 * float(g["demb_norm"])
    assert int((demb.abs().sum(1) > 0).sum()) == int(g["demb_nonzero"])


@pytest.mark.parametrize("M,N,D,seed", [(1, 1, 8, 0), (257, 1000, 8, 1), (4096, 8192, 8, 2), (100, 3000, 4, 3), (64, 500, 16, 4)])
def test_vq_quantize_matches_oracle_on_fresh_data(gpu_device, M, N, D, seed):
    from ldm_image_generator_amd import ops
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(M, D, generator=gen) * 1.5
    emb = torch.randn(N, D, generator=gen)
    if N > 10:
        emb[N - 1] = emb[3]                         # duplicate rows: the first index must win
        x[0] = emb[3]
    idx = ops.vq_quantize(x.cuda(), emb.cuda()).cpu().numpy()
    assert np.array_equal(idx, V.quantize(x.numpy(), emb.numpy()))


def test_vq_nan_query_and_ragged_sizes(gpu_device):
    from ldm_image_generator_amd import ops
    emb = torch.tensor([[0, 0, 0, 0, 0, 0, 0, 1.0], [1, 0, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0, 0]])
    x = torch.tensor([[1, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0.9], [float("nan"), 0, 0, 0, 0, 0, 0, 0]])
    assert ops.vq_quantize(x.cuda(), emb.cuda()).cpu().tolist() == [1, 0, 0]


def test_vae_calclate_loss_forward_matches_reference(gpu_device):
    """vae.py:36-43 forward (encoder -> + noise -> VQ loss, decoder -> L1) on the tiny VAE of the golden, noise replayed."""
    from ldm_image_generator_amd import synth
    from ldm_image_generator_amd.vae import VAE, Decoder, Encoder, VectorQuantizer
    g = load_golden("vq")
    enc = Encoder(channels=[32, 64, 32], stages=[1, 2, 1])
    enc.load_state_dict(synth.fill_state_dict(enc.state_dict()))
    dec = Decoder(channels=[64, 32, 32], stages=[1, 2, 1])
    dec.load_state_dict(synth.fill_state_dict(dec.state_dict()))
    vq = VectorQuantizer(num_embeddings=512, dim=8)
    with torch.no_grad():
        vq.embeddings.copy_(T(g["vae_emb"]))
    vae = VAE(enc, dec, vq).cuda()
    x = T(g["vae_x"]).cuda()
    noise = T(g["vae_noise"]).cuda()
    real_randn = torch.randn
    try:
        torch.randn = lambda *a, **k: noise                                     # replay the reference's latent noise
        with torch.no_grad():
            loss_recon, loss_reg, y = vae.calclate_loss(x, noise_gain=0.1)
    finally:
        torch.randn = real_randn
    assert rel_l2(y.cpu(), T(g["vae_y"])) < 1e-5
    assert abs(float(loss_recon) - float(g["vae_loss_recon"])) < 1e-5 * float(g["vae_loss_recon"])
    assert abs(float(loss_reg) - float(g["vae_loss_reg"])) < 1e-4 * float(g["vae_loss_reg"])
