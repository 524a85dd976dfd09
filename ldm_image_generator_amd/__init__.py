"""MI355X-native latent-diffusion hot path."""
