"""TEST INFRASTRUCTURE (oracle/): builds the REFERENCE's own ER-NeRF extensions for gfx950 so the GPU parity tests can run the reference's
kernels themselves beside the HIP path (raymarching.cu / gridencoder.cu / shencoder.cu / freqencoder.cu were "unpinned").

The sources are compiled where they lie under /root/reference/ernerf/*/src (never copied into the repo): PyTorch-ROCm's own hipify pass
(torch.utils.hipify, the tool `torch.utils.cpp_extension` runs on every CUDA extension) translates them into a scratch directory under
oracle/_ref/build, hipcc builds them against the real torch headers, and only the four extension modules stay in oracle/_ref/
(git-ignored; they travel to the GPU box like the product's own .so).  Nothing here is used by the product.

    python oracle/build_ref_ernerf.py          # no GPU needed (cross-compiles for gfx950)
"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = os.environ.get("MF_REFERENCE", "/root/reference")
# module name the reference's wrappers import (raymarching.py:9-12, sphere_harmonics.py:9-12, freq.py:9-12).
# NOT buildable here: gridencoder.cu -- its training-only backward kernel calls atomicAdd(__half2*, __half2) (gridencoder.cu:~300), an overload
# ROCm 7.2's hip_fp16.h does not provide; supplying one would be a stand-in for an API the image lacks, so the grid encoder stays pinned by the
# C restatement (oracle/ernerf_ref.c) + KATs only.
EXTS = {
    "_raymarching_face": ("raymarching", ["raymarching.cu", "bindings.cpp"]),
    "_shencoder": ("shencoder", ["shencoder.cu", "bindings.cpp"]),
    "_freqencoder": ("freqencoder", ["freqencoder.cu", "bindings.cpp"]),
}


# the reference's nvcc flags (raymarching/backend.py:6-9: -O3 -std=c++17 -U__CUDA_NO_HALF_OPERATORS__ -U__CUDA_NO_HALF_CONVERSIONS__
# -U__CUDA_NO_HALF2_OPERATORS__), spelled for hipcc: torch defines the __HIP_NO_HALF_* pair for every extension
HIPCC_FLAGS = ["-O3", "-std=c++17", "-U__HIP_NO_HALF_OPERATORS__", "-U__HIP_NO_HALF_CONVERSIONS__"]


def built():
    return all(any(f.startswith(name) and f.endswith(".so") for f in os.listdir(OUT)) for name in EXTS) if os.path.isdir(OUT) else False


def build(verbose=False):
    if built():
        return True
    if not os.path.isdir(os.path.join(REF, "ernerf")):
        return False                                   # GPU box: only the prebuilt modules exist
    os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
    os.environ.setdefault("MAX_JOBS", "4")
    from torch.utils.hipify import hipify_python
    from torch.utils import cpp_extension as ce
    os.makedirs(OUT, exist_ok=True)
    scratch = os.path.join(OUT, "build")
    for name, (pkg, files) in EXTS.items():
        src = os.path.join(REF, "ernerf", pkg, "src")
        work = os.path.join(scratch, pkg)
        shutil.rmtree(work, ignore_errors=True)
        res = hipify_python.hipify(project_directory=src, output_directory=work, includes=[os.path.join(src, "*")],
                                   extensions=(".cu", ".cuh", ".cpp", ".h", ".hpp"), is_pytorch_extension=True, show_detailed=verbose, show_progress=verbose)
        hip_src = []
        for f in files:
            r = res.get(os.path.join(src, f)) or res.get(os.path.join(work, f))
            hip_src.append(r.hipified_path if r is not None and r.hipified_path else os.path.join(work, f))
        bdir = os.path.join(work, "obj")
        os.makedirs(bdir, exist_ok=True)
        # same flags as the reference's backend.py (raymarching/backend.py:6-9) minus the nvcc-only switches
        # (torch.utils.cpp_extension.load would run hipify a second time on the translated files; this is the step it performs after hipify)
        ce._write_ninja_file_and_build_library(name=name, sources=hip_src, extra_cflags=["-O3", "-std=c++17"], extra_cuda_cflags=HIPCC_FLAGS,
                                               extra_sycl_cflags=[], extra_ldflags=[], extra_include_paths=[work], build_directory=bdir,
                                               verbose=verbose, with_cuda=True, with_sycl=False)
        for f in os.listdir(bdir):
            if f.endswith(".so"):
                shutil.copy2(os.path.join(bdir, f), os.path.join(OUT, f))
    shutil.rmtree(scratch, ignore_errors=True)           # translated sources are build by-products: only the modules stay
    return built()


if __name__ == "__main__":
    ok = build(verbose="-v" in sys.argv)
    print("[oracle/_ref] ER-NeRF reference extensions:", "built" if ok else "NOT built", sorted(os.listdir(OUT)) if os.path.isdir(OUT) else [])
    sys.exit(0 if ok else 1)
