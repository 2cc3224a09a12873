"""`wav2lip` as the reference imports it (lipasr.py:10 `from wav2lip import audio`), resolved to the
MI355X implementation.  Put this directory's parent (`mere-fusion_amd/dropin`) and the repository root
ahead of the reference checkout on sys.path; see INTEGRATION.md.  Modules that are not on the hot path
(`wav2lip.hparams`, `wav2lip.genavatar`, `wav2lip.face_detection`) fall through to the reference's directory."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
from mere_fusion_amd import placement as _placement  # noqa: E402

_placement.ensure_placed(session=False)      # multi-GPU node: the process takes a GPU before anything touches the device (lipreal.py:29 `device = 'cuda'`); the models charge a session each
from . import audio  # noqa: E402,F401
