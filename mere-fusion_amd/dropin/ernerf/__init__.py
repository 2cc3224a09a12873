"""`ernerf` as the reference imports it (app.py:15-17), resolved to the MI355X implementation for `nerf_triplane.network` and FALLING THROUGH to the
reference's own directory for everything else (`ernerf.nerf_triplane.{provider,utils,renderer,asr}`, `ernerf.encoding`, `ernerf.raymarching`, ...):
`pkgutil.extend_path` appends every other `ernerf/` directory found on sys.path, so this regular package does not shadow them."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
from mere_fusion_amd import placement as _placement  # noqa: E402

_placement.ensure_placed(session=False)      # multi-GPU node: this process takes its GPU before anything touches the device (app.py:378 `torch.device('cuda')`)
