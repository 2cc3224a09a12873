"""Drop-in for the reference's `wav2lip` package, hot-path subset only.

lipreal.py:25 does `from wav2lip.models import Wav2Lip`, lipasr.py:10 does
`from wav2lip import audio`.  Put the parent directory of THIS package ahead of the reference
checkout on sys.path and both imports resolve here, leaving lipreal.py,
lipasr.py, basereal.py, baseasr.py, app.py and webrtc.py untouched.
"""
