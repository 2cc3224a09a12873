#!/usr/bin/env python3
"""Disassembles every gfx950 code object inside libmerefusion_hip.so (no GPU needed) and counts the instruction form behind round 5's "wrong channel now and then":

    a packed-fp32 VALU instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) whose op_sel selects the HIGH register of src1 for the LOW result
    -- e.g. `v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0]` -- returns a wrong low half in lanes 48..63 (src1 read as zero) when another wave of the
    same SIMD is issuing MFMAs: tools/pkfma_repro.hip reproduces it in isolation on MI355X, 0.09 % of executions (gpurun_out -> profiles/r06_pkfma_study.txt).

src0 / src2 selects and every op_sel_hi form measured clean.  The compiler forms these instructions on its own (SLP-vectorised scalar code whose two multipliers
sit in one register pair), so the library is checked after every build: `python tools/isa_scan.py [lib.so]` prints one line per kernel that holds the form and exits 1
if any does; mere_fusion_amd/build.py runs it after linking and tests/test_isa_scan.py in the CPU suite."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
PK = re.compile(r"^\s*(v_pk_(?:fma|mul|add)_f32)\b(.*)$")
OPSEL = re.compile(r"\bop_sel:\[([01](?:,[01])*)\]")
FUNC = re.compile(r"^[0-9a-f]+ <([^>]+)>:")


def code_objects(lib):
    """the gfx950 code objects of every bundle in the library's .hip_fatbin section (one bundle per translation unit)"""
    with tempfile.TemporaryDirectory() as td:
        sec = os.path.join(td, "fatbin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, sec], check=True)
        blob = open(sec, "rb").read()
    out, pos = [], blob.find(MAGIC)
    while pos >= 0:
        n = int.from_bytes(blob[pos + 24:pos + 32], "little")
        q = pos + 32
        for _ in range(n):
            off, size, tlen = (int.from_bytes(blob[q + 8 * i:q + 8 * i + 8], "little") for i in range(3))
            triple = blob[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if "gfx950" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos = blob.find(MAGIC, pos + 1)
    return out


def scan(lib):
    """-> (offenders {kernel: [instruction, ...]}, census {(mnemonic, op_sel string or ''): count}, kernels scanned)"""
    offenders, census, kernels = {}, {}, 0
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f.name], check=True, capture_output=True, text=True).stdout
        fn = "?"
        for line in dis.splitlines():
            m = FUNC.match(line)
            if m:
                fn = m.group(1)
                kernels += 1
                continue
            m = PK.match(line)
            if not m:
                continue
            sel = OPSEL.search(m.group(2))
            bits = sel.group(1) if sel else ""
            census[(m.group(1), bits)] = census.get((m.group(1), bits), 0) + 1
            if bits and len(bits.split(",")) >= 2 and bits.split(",")[1] == "1":            # src1's HIGH register feeds the LOW result
                offenders.setdefault(fn, []).append(line.split("//")[0].strip())
    return offenders, census, kernels


def scratch_users(lib):
    """-> {kernel symbol: (scratch bytes per lane, vgpr count)} of the kernels whose code object asks for private (scratch) memory.  Such a kernel costs more to
    dispatch and keeps its spills / private arrays in memory ("Scratch in small kernels"): tests/test_isa_scan.py holds the list to the one
    kernel family that spills by design."""
    out = {}
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], check=True, capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            size = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
            vg = re.search(r"\.vgpr_count:\s+(\d+)", blk)
            if name and size and int(size.group(1)) > 0:
                out[name.group(1)] = (int(size.group(1)), int(vg.group(1)) if vg else -1)
    return out


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mere-fusion_amd", "libmerefusion_hip.so")
    offenders, census, kernels = scan(lib)
    total = sum(census.values())
    print(f"{os.path.normpath(lib)}: {kernels} functions, {total} packed-fp32 FMA / MUL / ADD instructions; by op_sel: "
          + ", ".join(f"{k[0]} op_sel:[{k[1]}] x {v}" for k, v in sorted(census.items()) if k[1]) + ("" if any(k[1] for k in census) else "none with op_sel"))
    sc = scratch_users(lib)
    print(f"  kernels with scratch: {len(sc)}" + ("" if not sc else " -- " + ", ".join(f"{k[:48]} {v[0]} B" for k, v in sorted(sc.items())[:8])))
    for fn, ins in sorted(offenders.items()):
        try:
            fn = subprocess.run(["c++filt", fn], capture_output=True, text=True).stdout.strip().split("(")[0]
        except OSError:
            pass
        print(f"  OFFENDER {fn}: {len(ins)} x e.g. {ins[0]}")
    return 1 if offenders else 0


if __name__ == "__main__":
    sys.exit(main())
