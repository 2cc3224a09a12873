"""`musetalk.models`: unet / vae resolve here; any other module falls through to the reference's directory."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
