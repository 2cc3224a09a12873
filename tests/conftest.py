import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Launch configurations: a forward never measures; it looks its implicit-GEMM layers up in the tuning table shipped beside the library
# (mere-fusion_amd/tune/gfx950.txt, the BASELINE.json shapes) and runs the cost model's pick for every other shape -- so the suite launches exactly what a
# deployment launches.  MF_AUTOTUNE=1 (the `autotuned` fixture) is the development mode that measures on the first forward at a batch size.


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture()
def autotuned(monkeypatch):
    """inside this test the first forward at a batch size measures its launch configurations (development mode of rounds 1-2)"""
    monkeypatch.setenv("MF_AUTOTUNE", "1")


@pytest.fixture(scope="session")
def lib_built():
    """The C-ABI library, built in-tree (hipcc cross-compiles without a GPU)."""
    from mere_fusion_amd import build
    return build.build(verbose=False)


@pytest.fixture(scope="session")
def sd0():
    from mere_fusion_amd import weights
    return weights.make_wav2lip_state_dict(0)


@pytest.fixture(scope="session")
def wav2lip_golden():
    return dict(np.load(os.path.join(GOLDEN, "wav2lip_golden.npz")))


@pytest.fixture(scope="session")
def conv_golden():
    return dict(np.load(os.path.join(GOLDEN, "conv_golden.npz")))


@pytest.fixture(scope="session")
def gpu_model_factory(lib_built, sd0):
    """Builds drop-in Wav2Lip modules on cuda:0, cached per precision."""
    cache = {}

    def make(precision):
        if precision not in cache:
            from mere_fusion_amd.wav2lip.models import Wav2Lip
            m = Wav2Lip(precision=precision)
            m.load_state_dict(sd0)
            cache[precision] = m.to("cuda").eval()
        return cache[precision]

    return make
