// Measurement seam: what the matrix pipes of THIS chip deliver to a kernel that does nothing but MFMAs, for the instruction mix one product costs in
// each operand format (bench.py reports it beside the nominal roofline peak; tools/mx_cross_probe.hip is the standalone version with the operand
// layout checks).  Per 128-deep slab of a 16 x 16 accumulator tile:
//   mix 0  bf16x3 (what ships): 12 x v_mfma_f32_16x16x32_bf16
//   mix 1  f16 + FP8 corrections: 4 x v_mfma_f32_16x16x32_f16 + 2 x v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3)
//   mix 2  f16 + FP6 corrections: 4 x v_mfma_f32_16x16x32_f16 + 2 x v_mfma_scale_f32_16x16x128_f8f6f4 (e2m3)
// Random (not zero) operand bits: the chip clocks to its power budget and zeros would flatter the number (MI355X_MICROARCH.md, DVFS give-back).
#include "mf_common.h"
#include <vector>

typedef __attribute__((ext_vector_type(8))) int pi32x8;
typedef __attribute__((ext_vector_type(4))) float pf32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 pf16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 pbf16x8;

namespace {
template <int MODE>
__global__ __launch_bounds__(256) void k_mfma_rate(const pi32x8* src, pf32x4* out, int iters) {
    const pi32x8 ra = src[threadIdx.x & 63], rb = src[64 + (threadIdx.x & 63)];
    const pbf16x8 ba = __builtin_bit_cast(pbf16x8, __builtin_shufflevector(ra, ra, 0, 1, 2, 3)), bb = __builtin_bit_cast(pbf16x8, __builtin_shufflevector(rb, rb, 0, 1, 2, 3));
    const pf16x8 ha = __builtin_bit_cast(pf16x8, __builtin_shufflevector(ra, ra, 0, 1, 2, 3)), hb = __builtin_bit_cast(pf16x8, __builtin_shufflevector(rb, rb, 0, 1, 2, 3));
    pf32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = pf32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 12; ++r)
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc[t], 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[t], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    acc[t] = MODE == 1 ? __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ra, rb, acc[t], 0, 0, 0, 127, 0, 127)
                                       : __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ra, rb, acc[t], 2, 2, 0, 127, 0, 127);
        }
    }
    pf32x4 s = acc[0];
#pragma unroll
    for (int t = 1; t < 8; ++t) s += acc[t];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
}  // namespace

extern "C" int mf_probe_mfma_ceiling(int mix, float* tflops_algorithmic) {
    MF_REQUIRE(mix >= 0 && mix <= 2 && tflops_algorithmic, "probe_mfma_ceiling: mix 0 (bf16x3), 1 (f16 + MX-fp8), 2 (f16 + MX-fp6)");
    const int wgs = 512, iters = 2000;
    pi32x8* src = nullptr; pf32x4* out = nullptr;
    MF_HIP(hipMalloc(&src, 128 * 32)); MF_HIP(hipMalloc(&out, (size_t)wgs * 256 * 16));
    std::vector<uint32_t> h(128 * 8);
    uint32_t x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((x >> 4) & 0x3f3f3f3fu) | 0x38003800u; }   // finite, moderate values in every format
    MF_HIP(hipMemcpy(src, h.data(), 128 * 32, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    MF_HIP(hipEventCreate(&e0)); MF_HIP(hipEventCreate(&e1));
    auto launch = [&](int n) {
        if (mix == 0) hipLaunchKernelGGL(k_mfma_rate<0>, dim3(wgs), dim3(256), 0, 0, src, out, n);
        else if (mix == 1) hipLaunchKernelGGL(k_mfma_rate<1>, dim3(wgs), dim3(256), 0, 0, src, out, n);
        else hipLaunchKernelGGL(k_mfma_rate<2>, dim3(wgs), dim3(256), 0, 0, src, out, n);
    };
    launch(200);
    MF_HIP(hipEventRecord(e0, 0));
    launch(iters);
    MF_HIP(hipEventRecord(e1, 0));
    MF_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    MF_HIP(hipEventElapsedTime(&ms, e0, e1));
    // every iteration advances 8 accumulator tiles by one 128-deep slab: 8 x 2 * 16 * 16 * 128 FLOP of the convolution's own arithmetic per wave
    *tflops_algorithmic = (float)((double)wgs * 4 * iters * 8 * 2.0 * 16 * 16 * 128 / (ms * 1e-3) / 1e12);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(src); (void)hipFree(out);
    return MF_OK;
}
