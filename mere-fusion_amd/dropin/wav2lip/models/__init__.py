"""lipreal.py:25 `from wav2lip.models import Wav2Lip`; the training-time discriminators (`wav2lip.models.syncnet`) stay the reference's."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
from mere_fusion_amd.wav2lip.models import Wav2Lip  # noqa: E402,F401
