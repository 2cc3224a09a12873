#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;
template <int MODE> __global__ __launch_bounds__(256) void k(const i32x4* src, float* out, int iters) {
    const bf16x8 a = __builtin_bit_cast(bf16x8, src[threadIdx.x & 63]), b = __builtin_bit_cast(bf16x8, src[64 + (threadIdx.x & 63)]);
    if (MODE == 0) {
        f32x4 acc[8];
        for (int t = 0; t < 8; ++t) acc[t] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 12; ++r)
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
        f32x4 s = acc[0]; for (int t = 1; t < 8; ++t) s += acc[t];
        out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    } else {
        // same FLOPs per iteration: 8 tiles of 16x16 x 128K x 3 passes = 2 tiles of 32x32 x 128K x 3 = 2 x 8 x 3 = 48 MFMAs of 32x32x16
        f32x16 acc[2];
        for (int t = 0; t < 2; ++t) for (int e = 0; e < 16; ++e) acc[t][e] = 0;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 24; ++r)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
        float s = 0; for (int t = 0; t < 2; ++t) for (int e = 0; e < 16; ++e) s += acc[t][e];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}
int main() {
    i32x4* src; float* out; const int wgs = 512, iters = 2000;
    hipMalloc(&src, 128 * 16); hipMalloc(&out, wgs * 256 * 4);
    std::vector<uint32_t> h(128 * 4); uint32_t x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((x >> 4) & 0x3f3f3f3fu) | 0x38003800u; }
    hipMemcpy(src, h.data(), 128 * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, src, out, 200); else hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, src, out, 200);
        hipEventRecord(e0, 0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, src, out, iters); else hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, src, out, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)wgs * 4 * iters * 8 * 2.0 * 16 * 16 * 128;
        printf("%s: %.3f ms -> %.1f TF bf16x3-algorithmic (%.0f TF raw bf16)\n", mode ? "32x32x16" : "16x16x32", ms, flop / (ms * 1e-3) / 1e12, 3 * flop / (ms * 1e-3) / 1e12);
    }
}
