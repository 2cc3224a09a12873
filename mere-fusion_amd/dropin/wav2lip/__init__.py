"""`wav2lip` as the reference imports it (lipasr.py:10 `from wav2lip import audio`), resolved to the
MI355X implementation.  Put this directory's parent (`mere-fusion_amd/dropin`) and the repository root
ahead of the reference checkout on sys.path; see INTEGRATION.md."""
from mere_fusion_amd.wav2lip import audio  # noqa: F401
