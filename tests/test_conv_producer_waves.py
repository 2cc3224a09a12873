"""The producer-wave operand path of k_conv_igemm (LD 3: four DMA-only waves feed a ring of LDS stages; ld 3 = 64-deep stages, ld 4 = 32-deep ones).
The tuning table selects it per layer; here it is FORCED onto every layer it can run (MF_FORCE_LD, read once per process: hence child processes) and the
conv goldens recorded from the reference's own modules, the Wav2Lip generator golden, the reduced-config MuseTalk parity tests and the GroupNorm-statistics
epilogue must hold unchanged -- 64-deep stages (ld 3) and 32-deep ones on the 64 x 64 tile (ld 4)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra, targets, timeout=1500):
    env = dict(os.environ, **env_extra)
    env.pop("MF_TUNE_CACHE", None)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + targets, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1500:])
    return r.stdout


def test_forced_on_every_eligible_layer_goldens_hold(lib_built):
    out = _run({"MF_FORCE_LD": "3"}, ["tests/test_wav2lip_gpu.py::test_conv_geometry_vs_golden", "tests/test_wav2lip_gpu.py::test_conv_geometry_ragged_batch",
                                      "tests/test_wav2lip_gpu.py::test_generator_vs_reference_golden", "tests/test_wav2lip_gpu.py::test_graph_replay_is_deterministic",
                                      "tests/test_musetalk.py", "tests/test_conv_wide.py::test_conv_leaves_groupnorm_statistics_of_its_output"])
    assert " passed" in out


@pytest.mark.parametrize("tile,split,ld", [("64x64", "3", "4"), ("128x128", "2", "3"), ("128x128", "3", "4"), ("128x64", "1", "3"), ("128x64", "2", "4"), ("64x64", "16", "3")])
def test_forced_tiles_and_splits(lib_built, tile, split, ld):
    """every tile of the path, shallow and deep split-K (a workgroup with fewer K steps than the ring has stages included), against the conv goldens + the UNet"""
    out = _run({"MF_FORCE_LD": ld, "MF_FORCE_TILE": tile, "MF_FORCE_SPLIT": split},
               ["tests/test_wav2lip_gpu.py::test_conv_geometry_vs_golden", "tests/test_musetalk.py::test_hip_unet_vs_oracle", "tests/test_musetalk.py::test_hip_geglu_projection"])
    assert " passed" in out


@pytest.mark.parametrize("split,ld", [("1", "3"), ("1", "4"), ("3", "3")])
def test_forced_128x80_tile(lib_built, split, ld):
    """the 80-channel tile (five channel fragments per wave, weight pieces that do not divide among the four producers): outputs against fp64 and the GroupNorm
    statistics epilogue at 10 / 20 / 40 channels per group; layers it cannot take (channels not a multiple of 80) fall back to the 64 x 64 tile"""
    out = _run({"MF_FORCE_LD": ld, "MF_FORCE_TILE": "128x80", "MF_FORCE_SPLIT": split},
               ["tests/test_conv_wide.py::test_channel_multiples_of_80_vs_fp64", "tests/test_conv_wide.py::test_conv_leaves_groupnorm_statistics_of_its_output",
                "tests/test_musetalk.py::test_hip_unet_vs_oracle"])
    assert " passed" in out
