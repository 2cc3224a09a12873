"""lipreal.py:25 `from wav2lip.models import Wav2Lip`."""
from mere_fusion_amd.wav2lip.models import Wav2Lip  # noqa: F401
