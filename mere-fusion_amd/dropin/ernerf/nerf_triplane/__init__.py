"""`ernerf.nerf_triplane`: `network` resolves here; `renderer`, `provider`, `utils`, `asr`, `gui` fall through to the reference's directory."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
