"""MI355X-native latent-diffusion hot path (UNet denoise loop + VAE decode).

Public surface mirrors uthree/ldm-image-generator's modules:
    from ldm_image_generator_amd.unet import UNet
    from ldm_image_generator_amd.vae import Decoder, VAE
    from ldm_image_generator_amd.ddpm import DDPM
All arithmetic runs in hand-written gfx950 HIP kernels behind the C ABI declared
in include/ldm_hip.h (libldm_hip.so).  There is no CPU fallback.
"""
__version__ = "0.1.0"
