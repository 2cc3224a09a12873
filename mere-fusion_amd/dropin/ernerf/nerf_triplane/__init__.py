"""`ernerf.nerf_triplane`: `network` and `utils` resolve here (each loads the reference's own module and puts the MI355X pieces in front); `renderer`, `provider`,
`asr`, `gui` fall through to the reference's directory."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
