"""`musetalk.utils`: `utils` and `blending` resolve here; `preprocessing`, `face_parsing`, `face_detection`, `dwpose` fall through
to the reference's directory.  The reference's own `musetalk/utils/__init__.py:1-5` appends `<musetalk>/utils` to sys.path (so that
`blending.py` can `from face_parsing import FaceParsing`); that side effect is reproduced for every directory the package spans."""
import sys
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
for _p in list(__path__)[1:]:
    if _p not in sys.path:
        sys.path.append(_p)
