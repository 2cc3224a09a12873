"""`musetalk.utils.utils` drop-in, hot-path subset: the two loaders musereal.py:57 / app.py call
(musetalk/utils/utils.py:66-75)."""
from ..models.unet import UNet
from ..models.vae import VAE
from ..whisper.audio2feature import Audio2Feature


def load_audio_model():
    return Audio2Feature(model_path="./models/whisper/tiny.pt")


def load_diffusion_model():
    vae = VAE(model_path="./models/sd-vae-ft-mse/")
    unet = UNet(unet_config="./models/musetalk/musetalk.json", model_path="./models/musetalk/pytorch_model.bin")
    return vae, unet, unet.pe
