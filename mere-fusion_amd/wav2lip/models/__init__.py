from .generator import Wav2Lip  # noqa: F401  (same import path as wav2lip/models/__init__.py:1)
