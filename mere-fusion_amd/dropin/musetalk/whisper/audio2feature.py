"""museasr.py:8 `from musetalk.whisper.audio2feature import Audio2Feature`."""
from mere_fusion_amd.musetalk.whisper.audio2feature import Audio2Feature  # noqa: F401
