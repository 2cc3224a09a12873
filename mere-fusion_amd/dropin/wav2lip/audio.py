"""lipasr.py:10,23 `from wav2lip import audio; audio.melspectrogram(wav)`."""
from mere_fusion_amd.wav2lip.audio import *  # noqa: F401,F403
from mere_fusion_amd.wav2lip.audio import melspectrogram  # noqa: F401
