"""`musetalk.utils.utils` (musetalk/utils/utils.py:19-75): every name musereal.py:21-24 / app.py import from it."""
from mere_fusion_amd.musetalk.utils.utils import (  # noqa: F401
    datagen, get_file_type, get_video_fps, load_all_model, load_audio_model, load_diffusion_model)
