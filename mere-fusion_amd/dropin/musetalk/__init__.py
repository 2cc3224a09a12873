"""`musetalk` as the reference imports it (museasr.py:8), resolved to the MI355X implementation for the hot-path modules
(`models.unet`, `models.vae`, `whisper.audio2feature`, `utils.utils`, `utils.blending`) and FALLING THROUGH to the reference's own
package for everything else (`musetalk.mere_musetalk`, `musetalk.utils.preprocessing`, `musetalk.whisper.whisper`, ...):
`pkgutil.extend_path` appends every other `musetalk/` directory found on sys.path, so this regular package does not shadow them."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
from mere_fusion_amd import placement as _placement  # noqa: E402

_placement.ensure_placed(session=False)      # multi-GPU node: the process takes a GPU before anything touches the device (musereal.py:58); the models charge a session each
