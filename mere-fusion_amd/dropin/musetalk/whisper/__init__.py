"""`musetalk.whisper`: audio2feature resolves here; the vendored `musetalk.whisper.whisper` package stays the reference's."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
