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

# The library measures its implicit-GEMM launch configurations on the first forward of every (handle, batch size) -- seconds per full-size network.
# The suite builds dozens of handles, so it runs with the cost model alone unless a test asks for the production default (the `autotuned`
# fixture: tests/test_musetalk_full.py, tests/test_autotune.py).  Handles read the variable when they are created.
os.environ.setdefault("MF_AUTOTUNE", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture()
def autotuned(monkeypatch):
    """handles created inside this test use the production default: measured launch configurations"""
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
