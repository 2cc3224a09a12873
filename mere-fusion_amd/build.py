"""Builds libmerefusion_hip.so (hipcc, gfx950 only) in-tree, next to this file.

No GPU is needed to build: hipcc cross-compiles.  The .so is git-ignored but travels to the GPU
box with the repo snapshot.  Every source is compiled to its own object (in parallel, only when it or a
header changed) under build/obj, then linked: a one-kernel edit rebuilds in seconds instead of minutes.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmerefusion_hip.so")
OBJ = os.path.normpath(os.path.join(HERE, "..", "build", "obj"))
SOURCES = ["mf_api.cpp", "mf_conv.hip", "mf_conv_halo.hip", "mf_conv_halo2.hip", "mf_conv_thin.hip", "mf_conv_tail.hip", "mf_aux.hip", "mf_wav2lip.hip", "mf_conv_api.hip", "mf_mel.hip",
           "mf_nn.hip", "mf_attn.hip", "mf_whisper.hip", "mf_musetalk.hip", "mf_nerf.hip", "mf_nerf_net.hip", "mf_nerf_fused.hip", "mf_nerf_torso.hip", "mf_nerf_audio.hip",
           "mf_blend.hip", "mf_session.hip", "mf_wav2vec2.hip", "mf_net.hip", "mf_probe.hip"]
# -pragma-unroll-threshold: the 256 x 256 implicit-GEMM tile's `#pragma unroll` loops (128 accumulator registers, 8 x 4 fragments) exceed LLVM's default
# cap of 16384 for pragma-driven unrolling; a loop left rolled indexes the accumulator array dynamically, which puts it in scratch (592 bytes per lane,
# the kernel 5 x slower: found in round 3 when unrelated code left the kernel and the estimate tipped over)
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-mllvm", "-pragma-unroll-threshold=262144"]


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [
        os.path.join(HERE, "..", "include", "merefusion.h"), os.path.abspath(__file__)]


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + _headers()
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in _headers())
    jobs = []
    for s in _sources():
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append([hipcc] + CFLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[mere-fusion_amd] " + " ".join(cmd[-3:]), file=sys.stderr)
        subprocess.run(cmd, check=True)

    workers = max(1, min(len(jobs), int(os.environ.get("MF_BUILD_JOBS", min(8, os.cpu_count() or 1)))))
    if jobs:
        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(run, jobs))
    link = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + [os.path.join(OBJ, s + ".o") for s in _sources()]
    if verbose:
        print(f"[mere-fusion_amd] link {LIB} ({len(jobs)} objects rebuilt)", file=sys.stderr)
    subprocess.run(link, check=True)
    isa_check(verbose)
    return LIB


def isa_check(verbose=True):
    """The gfx950 packed-fp32 erratum (csrc/mf_common.h `mf_opaque`): the freshly linked library must not contain the instruction form anywhere --
    the compiler produces it on its own, so every build is disassembled and checked (3 s).  A build that fails here is removed: it must not travel to a GPU box."""
    sys.path.insert(0, os.path.normpath(os.path.join(HERE, "..", "tools")))
    try:
        import isa_scan
    finally:
        sys.path.pop(0)
    offenders, census, kernels = isa_scan.scan(LIB)
    if verbose:
        print(f"[mere-fusion_amd] ISA scan: {kernels} functions, {sum(census.values())} packed-fp32 instructions, {sum(len(v) for v in offenders.values())} of the erratum form", file=sys.stderr)
    if offenders:
        os.replace(LIB, LIB + ".rejected")
        lines = "\n".join(f"  {fn}: {len(ins)} x {ins[0]}" for fn, ins in sorted(offenders.items()))
        raise RuntimeError("libmerefusion_hip.so contains packed-fp32 instructions whose LOW result takes src1's HIGH register (wrong results beside MFMAs on gfx950: "
                           "tools/pkfma_repro.hip).  Break the pairing at the source with mf_opaque() (csrc/mf_common.h):\n" + lines)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
