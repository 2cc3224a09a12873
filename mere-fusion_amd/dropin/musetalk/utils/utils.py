from mere_fusion_amd.musetalk.utils.utils import load_audio_model, load_diffusion_model  # noqa: F401
