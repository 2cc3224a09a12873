#!/usr/bin/env python3
"""Second step of the packed-FMA root-cause study: builds of the library whose ONLY difference is how k_conv_igemm's LayerNorm-folding epilogue
computes rstd * (acc - mean * colsum) + bias' (k_splitk_epilogue keeps the shipped pinned scalar form), written as explicit inline asm so that the instruction FORM and
its TIMING can be varied independently:
  asm_bcast      v_pk_fma_f32 with op_sel_hi:[1,0,1]: mean / rstd in the LOW half of a register pair whose HIGH half holds an unrelated live value -- the form the
                 compiler produced in the failing build (control: expected to fail)
  asm_bcast_nop  the same instructions, each followed by s_nop 7 (form kept, timing relaxed)
  asm_dup        v_pk_fma_f32 WITHOUT the broadcast select: mean / rstd copied into both halves of the pair (v_mov), default op_sel
  asm_bcast_same the broadcast form, but the high half of the pair ALSO holds mean / rstd (if the select is sometimes ignored the result is still right)
  asm_selhi      (mean, rstd) in one pair; the rstd multiply as v_pk_fma_f32 ... op_sel:[0,1,0] (the LOW result takes the HIGH register): the form the compiler used
                 for the last fragment of a wave -- the only instructions whose results were wrong in the failing build (tools/pkfma_dump_analyze.py)
  asm_selhi_nop  the same with s_nop 7 after each
    python tools/pkfma_variants2.py      (here; build_ab/lib<variant>.so travel to the GPU box; tools/pkfma_study.sh runs them)"""
import os, subprocess, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.chdir(ROOT)
src = open("mere-fusion_amd/csrc/mf_conv.hip").read()
a = src.index("                float t0 = acc[i][j][0] - mu * cs.x, t1 = acc[i][j][1] - mu * cs.y")
b = src.index('                asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));', a)
b = src.index("\n", b) + 1
HEAD = '''                typedef float f2v __attribute__((ext_vector_type(2)));
                const f2 _unused = {0.f, 0.f}; (void)_unused;
'''
def body(hi_mu, hi_rs, sel, nop, sel2=None):
    n = "\\n\\ts_nop 7" if nop else ""
    sel2 = sel if sel2 is None else sel2
    return f'''                typedef float f2v __attribute__((ext_vector_type(2)));
                const f2v mup = {{mu, {hi_mu}}}, rsp = {{rs, {hi_rs}}};
                const f2v c01 = {{cs.x, cs.y}}, c23 = {{cs.z, cs.w}}, a01 = {{acc[i][j][0], acc[i][j][1]}}, a23 = {{acc[i][j][2], acc[i][j][3]}};
                const f2v b01 = {{bq[i].x, bq[i].y}}, b23 = {{bq[i].z, bq[i].w}};
                f2v t01, t23, o01, o23;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 {sel} neg_lo:[1,0,0] neg_hi:[1,0,0]{n}" : "=v"(t01) : "v"(c01), "v"(mup), "v"(a01));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 {sel} neg_lo:[1,0,0] neg_hi:[1,0,0]{n}" : "=v"(t23) : "v"(c23), "v"(mup), "v"(a23));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 {sel2}{n}" : "=v"(o01) : "v"(t01), "v"(rsp), "v"(b01));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 {sel2}{n}" : "=v"(o23) : "v"(t23), "v"(rsp), "v"(b23));
                v[0] = o01.x; v[1] = o01.y; v[2] = o23.x; v[3] = o23.y;
'''
junk_mu, junk_rs = "__int_as_float(m)", "__int_as_float(c + 0x3ff00000)"          # unrelated live values (a row index; the high word of an fp64 near 1.0)
variants = {
    "asm_bcast": body(junk_mu, junk_rs, "op_sel_hi:[1,0,1]", False),
    "asm_bcast_nop": body(junk_mu, junk_rs, "op_sel_hi:[1,0,1]", True),
    "asm_dup": body("mu", "rs", "", False),
    "asm_bcast_same": body("mu", "rs", "op_sel_hi:[1,0,1]", False),
    # (mean, rstd) in ONE pair, as the compiler kept them for the last fragment of a wave: the rstd multiply selects the HIGH register for both results
    "asm_selhi": body("rs", "rs", "op_sel_hi:[1,0,1]", False, sel2="op_sel:[0,1,0]").replace("rsp = {rs, rs}", "rsp = {mu, rs}"),
    "asm_selhi_nop": body("rs", "rs", "op_sel_hi:[1,0,1]", True, sel2="op_sel:[0,1,0]").replace("rsp = {rs, rs}", "rsp = {mu, rs}"),
}
import sys as _s
if len(_s.argv) > 1:
    variants = {k: v for k, v in variants.items() if k in _s.argv[1:]}
F = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=262144 -Imere-fusion_amd/csrc -Iinclude".split()
os.makedirs("build_ab", exist_ok=True)
procs = []
for name, code in variants.items():
    tmp = f"mere-fusion_amd/csrc/.ab_{name}.hip"
    open(tmp, "w").write(src[:a] + code + src[b:])
    procs.append((name, tmp, subprocess.Popen(["/opt/rocm/bin/hipcc"] + F + ["-c", tmp, "-o", f"build_ab/{name}.o"])))
objs = [os.path.join("build/obj", f) for f in sorted(os.listdir("build/obj")) if f.endswith(".o") and f != "mf_conv.hip.o"]
for name, tmp, p in procs:
    rc = p.wait()
    os.replace(tmp, f"build_ab/mf_conv_{name}.hip")
    if rc:
        sys.exit(f"{name}: compile failed")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", f"build_ab/lib{name}.so"] + objs + [f"build_ab/{name}.o"], check=True)
    print(f"build_ab/lib{name}.so")
