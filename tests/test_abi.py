"""CPU: the C-ABI library builds, loads, and exports every symbol include/merefusion.h declares.
No compute is attempted without a GPU; error paths that need no device are exercised."""
import ctypes as C
import os
import re

import pytest
import torch

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "merefusion.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mf_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib_built):
    from mere_fusion_amd import _lib
    lib = C.CDLL(lib_built)
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in merefusion.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    assert _lib.lib().mf_abi_version() == 4


def test_errors_without_device(lib_built):
    from mere_fusion_amd import _lib
    l = _lib.lib()
    assert l.mf_melspec_frames(16640) == 84 and l.mf_melspec_frames(7040) == 36 and l.mf_melspec_frames(48000) == 241
    if not torch.cuda.is_available():
        rc = l.mf_init(0)
        assert rc == -3 and b"no HIP device" in l.mf_last_error()
    # null handle -> MF_ERR_INVALID, never a crash
    assert l.mf_wav2lip_forward(None, None, None, None, 1, None) == -1
    assert b"null" in l.mf_last_error()
    assert l.mf_melspec(None, 10, None, 0, None) == -1 and l.mf_melspec(None, 0, None, 0, None) == -1


def test_product_refuses_cpu_tensors(lib_built, sd0):
    from mere_fusion_amd.wav2lip.models import Wav2Lip
    m = Wav2Lip()
    missing, unexpected = m.load_state_dict(sd0)
    assert not missing and not unexpected
    assert sorted(m.state_dict().keys()) == sorted(sd0.keys())
    m.eval()
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 1, 80, 16), torch.zeros(1, 6, 96, 96))
    m.train()
    with pytest.raises(RuntimeError, match="inference-only"):
        m(torch.zeros(1, 1, 80, 16), torch.zeros(1, 6, 96, 96))


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "mere-fusion_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "oracle/" not in src or f.endswith(".md"), f


def test_dropin_import_paths(lib_built):
    """The two imports the reference makes (lipreal.py:25, lipasr.py:10) resolve to this repository when
    mere-fusion_amd/dropin precedes the reference on sys.path."""
    import subprocess
    import sys
    code = ("import sys; sys.path[:0] = [%r, %r]; "
            "from wav2lip.models import Wav2Lip; from wav2lip import audio; "
            "import mere_fusion_amd.wav2lip.models as M; "
            "assert Wav2Lip is M.Wav2Lip and hasattr(audio, 'melspectrogram'); print('ok')"
            % (os.path.join(ROOT, "mere-fusion_amd", "dropin"), ROOT))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/")
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr
