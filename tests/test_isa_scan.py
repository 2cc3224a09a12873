"""The gfx950 packed-fp32 erratum (round 5's "wrong channel now and then"; DESIGN.md section 4, tools/pkfma_repro.hip): a v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32
whose op_sel takes src1's HIGH register for the LOW result returns a wrong low half in lanes 48..63 beside MFMAs.  The compiler forms that instruction on its own,
so the built library is disassembled (no GPU needed) and every kernel checked."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_library_has_no_packed_fp32_instruction_of_the_erratum_form(lib_built):
    import isa_scan
    offenders, census, kernels = isa_scan.scan(os.path.join(ROOT, "mere-fusion_amd", "libmerefusion_hip.so"))
    assert kernels >= 150 and sum(census.values()) > 1000, (kernels, sum(census.values()))      # the scan really saw the library's kernels and its packed arithmetic
    assert not offenders, {k: v[:2] for k, v in offenders.items()}


def test_scanner_flags_the_form_and_only_the_form(tmp_path):
    """The rule itself, on a hand-written kernel: src1's op_sel bit is the offender; src0 / src2 selects and op_sel_hi are not (all measured, pkfma_repro)."""
    import isa_scan
    src = tmp_path / "k.hip"
    src.write_text('''#include <hip/hip_runtime.h>
typedef float f2v __attribute__((ext_vector_type(2)));
__global__ void k_bad(f2v* p) { f2v a = p[0], b = p[1], c = p[2], r;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); p[3] = r;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b)); p[4] = r; }
__global__ void k_good(f2v* p) { f2v a = p[0], b = p[1], c = p[2], r;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); p[3] = r;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); p[4] = r;
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); p[5] = r; }
''')
    obj = tmp_path / "k.o"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-c", str(src), "-o", str(obj)], check=True, capture_output=True)
    offenders, census, kernels = isa_scan.scan(str(obj))
    assert list(offenders) == ["_Z5k_badPDv2_f"] and len(offenders["_Z5k_badPDv2_f"]) == 2, offenders
    assert census[("v_pk_fma_f32", "1,0,1")] == 1 and census[("v_pk_add_f32", "")] >= 1


def test_only_the_f16_fp6_halo_tile_uses_scratch(lib_built):
    """A kernel with scratch costs more to dispatch and keeps private arrays / spills in memory (round 6: the ER-NeRF audio encoder's staging array, 160 B per lane, and
    the loop's tail kernel, 24 B, each cost microseconds per frame).  The only kernels allowed to ask for it are the LDS-weights halo tiles, which spill a few registers
    at their 256-register cap by design (round 4)."""
    import isa_scan
    sc = isa_scan.scratch_users(os.path.join(ROOT, "mere-fusion_amd", "libmerefusion_hip.so"))
    others = {k: v for k, v in sc.items() if "k_conv3x3_halo_w" not in k}
    assert not others, others
    assert all(v[0] <= 128 for v in sc.values()), sc            # ... and only a few registers' worth
