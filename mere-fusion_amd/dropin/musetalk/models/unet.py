from mere_fusion_amd.musetalk.models.unet import UNet, PositionalEncoding  # noqa: F401
