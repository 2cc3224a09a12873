from mere_fusion_amd.musetalk.models.vae import VAE  # noqa: F401
