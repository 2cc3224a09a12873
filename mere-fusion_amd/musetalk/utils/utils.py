"""`musetalk.utils.utils` drop-in (musetalk/utils/utils.py:19-75): the loaders musereal.py:57 / app.py call and the small host
helpers musereal.py:21 imports beside them.  Same names, arguments and return tuples as the reference."""
import os

import numpy as np
import torch

from ... import placement
from ..models.unet import UNet
from ..models.vae import VAE
from ..whisper.audio2feature import Audio2Feature


def load_audio_model():
    """musetalk/utils/utils.py:66-68."""
    return Audio2Feature(model_path="./models/whisper/tiny.pt")


def load_diffusion_model():
    """musetalk/utils/utils.py:70-75 -> (vae, unet, pe); `pe` is the UNet's fused positional encoding marker."""
    vae = VAE(model_path="./models/sd-vae-ft-mse/")
    unet = UNet(unet_config="./models/musetalk/musetalk.json", model_path="./models/musetalk/pytorch_model.bin")
    return vae, unet, unet.pe


def load_all_model():
    """musetalk/utils/utils.py:19-25 -> (audio_processor, vae, unet, pe).  musereal.py:55 calls it in the session's own process (`inference`): on a multi-GPU
    node this is where that process takes its GPU (placement.py), before anything touches the device."""
    placed = placement.ensure_placed(session=True)
    audio_processor = load_audio_model()
    vae, unet, pe = load_diffusion_model()
    if placed is not None:
        import weakref
        weakref.finalize(unet, placement.uncharge, 1.0)               # the session's share of its GPU goes back when its models die
    return audio_processor, vae, unet, pe


def get_file_type(video_path):
    """musetalk/utils/utils.py:27-35: 'image' / 'video' / 'unsupported' by extension."""
    ext = os.path.splitext(video_path)[1].lower()
    if ext in (".jpg", ".jpeg", ".png", ".bmp", ".tif", ".tiff"):
        return "image"
    if ext in (".avi", ".mp4", ".mov", ".flv", ".mkv"):
        return "video"
    return "unsupported"


def get_video_fps(video_path):
    """musetalk/utils/utils.py:37-41 (avatar preparation only; needs OpenCV like the reference)."""
    import cv2
    video = cv2.VideoCapture(video_path)
    fps = video.get(cv2.CAP_PROP_FPS)
    video.release()
    return fps


def datagen(whisper_chunks, vae_encode_latents, batch_size=8, delay_frame=0):
    """musetalk/utils/utils.py:43-64: batches of (stacked whisper chunks, concatenated latents), the latents walked cyclically
    from `delay_frame`; the last batch may be short."""
    whisper_batch, latent_batch = [], []
    n = len(vae_encode_latents)
    for i, w in enumerate(whisper_chunks):
        whisper_batch.append(w)
        latent_batch.append(vae_encode_latents[(i + delay_frame) % n])
        if len(latent_batch) >= batch_size:
            yield np.stack(whisper_batch), torch.cat(latent_batch, dim=0)
            whisper_batch, latent_batch = [], []
    if latent_batch:
        yield np.stack(whisper_batch), torch.cat(latent_batch, dim=0)
