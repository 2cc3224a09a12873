# Builds of the library that differ ONLY in the LayerNorm-folding epilogues of mf_conv.hip (k_conv_igemm, k_splitk_epilogue): the root-cause study of the
# "wrong channel now and then" finding.   bash tools/pkfma_variants.sh   ->  build_ab/lib{packed,...}.so  (run here; the .so files travel to the GPU box)
#   packed      the asm pins removed: the compiler is free to form v_pk_fma_f32 with op_sel broadcasts (the failing form of round 5)
#   packed_nop  packed + an s_nop 7 between the fp64 statistics block and the first FMA and between the two FMA groups (timing only, same instructions)
#   packed_f32  packed, mean / rstd computed in fp32 (no DP-rate instructions in front of the packed FMAs)
#   packed_dup  packed, mu / rs passed through a v_mov each into BOTH halves of an aligned pair (no op_sel broadcast: the dead upper half is gone)
set -e
cd "$(dirname "$0")/.."
mkdir -p build_ab
SRC=mere-fusion_amd/csrc/mf_conv.hip
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=262144 -Imere-fusion_amd/csrc -Iinclude"
strip_pins() { sed -e '/asm volatile("" : "+v"(t0)/d' -e '/asm volatile("" : "+v"(v\[0\])/d' -e '/asm volatile("" : "+v"(o\[0\])/d' "$1"; }
strip_pins $SRC > build_ab/mf_conv_packed.hip
# s_nop variant: a scheduling barrier + nops right after rs is computed (both kernels)
sed -e 's|\(const float mu = (float)mean, rs = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)a.ln_eps));\)|\1 __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 7"); asm volatile("s_nop 7"); __builtin_amdgcn_sched_barrier(0);|' build_ab/mf_conv_packed.hip > build_ab/mf_conv_packed_nop.hip
sed -e 's|const float mu = (float)mean, rs = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)a.ln_eps));|const float mu = (float)sq.x * a.ln_inv_c, vf = (float)sq.y * a.ln_inv_c - mu * mu, rs = __frsqrt_rn((vf > 0.f ? vf : 0.f) + a.ln_eps); (void)mean; (void)var;|' build_ab/mf_conv_packed.hip > build_ab/mf_conv_packed_f32.hip
for v in packed packed_nop packed_f32; do
  grep -c "asm volatile(\"\"" build_ab/mf_conv_$v.hip || true
  ( cp build_ab/mf_conv_$v.hip mere-fusion_amd/csrc/.ab_$v.hip; /opt/rocm/bin/hipcc $F -c mere-fusion_amd/csrc/.ab_$v.hip -o build_ab/$v.o; rm -f mere-fusion_amd/csrc/.ab_$v.hip ) &
done
wait
objs=$(ls build/obj/*.o | grep -v "/mf_conv.hip.o")
for v in packed packed_nop packed_f32; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o build_ab/lib$v.so $objs build_ab/$v.o
  echo build_ab/lib$v.so
done
