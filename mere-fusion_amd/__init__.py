"""mere-fusion_amd: MI355X-native audio-to-face frame generator for mere-fusion's render loop.

Only the per-chunk inference path underneath lipreal.py / musereal.py lives here:
HIP kernels + a C ABI (csrc/, include/merefusion.h) and the Python host side that mirrors the
reference's own interfaces (`wav2lip.models.Wav2Lip`, `wav2lip.audio.melspectrogram`).
"""
__version__ = "0.1.0"
