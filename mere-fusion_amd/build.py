"""Builds libmerefusion_hip.so (hipcc, gfx950 only) in-tree, next to this file.

No GPU is needed to build: hipcc cross-compiles.  The .so is git-ignored but travels to the GPU
box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmerefusion_hip.so")
SOURCES = ["mf_api.cpp", "mf_conv.hip", "mf_conv_halo.hip", "mf_conv_halo2.hip", "mf_aux.hip", "mf_wav2lip.hip", "mf_conv_api.hip", "mf_mel.hip", "mf_nn.hip", "mf_attn.hip", "mf_whisper.hip", "mf_musetalk.hip", "mf_nerf.hip", "mf_nerf_net.hip", "mf_nerf_fused.hip", "mf_nerf_audio.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [
        os.path.join(HERE, "..", "include", "merefusion.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print("[mere-fusion_amd] " + " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
