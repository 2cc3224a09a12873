// Reproducer for the "wrong channel now and then" finding: the LayerNorm-folding epilogue -- fp64 statistics -> (float) mean / rstd ->
// rstd * (acc - mean * colsum) + bias over channel quads -- in isolation, in the two forms the compiler gives it:
//   PACKED: plain C++ (clang forms v_pk_fma_f32 with op_sel broadcasts of mean / rstd; check with llvm-objdump), SCALAR: the product's pinned scalar FMAs.
// Every workgroup computes the same rows from the same inputs; every wave's result is compared IN THE KERNEL with the value the host computed (the arithmetic
// is IEEE fp64 / fp32 FMA: exactly reproducible), mismatches counted by atomics.  Three more waves per SIMD run other instruction mixes (fp64 chains, MFMAs,
// LDS traffic, global loads) so that a wave's consecutive instructions issue back to back or not at random -- the condition under which a missing wait state
// shows as a timing-dependent error.      hipcc --offload-arch=gfx950 -O3 tools/pkfma_repro.hip -o tools/bin/pkfma_repro   (build here, run on the GPU box)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));

template <bool PACKED>
__device__ __forceinline__ void ln_quad(const float (&acc)[4], float mu, float rs, const float4 cs, const float4 bq, float (&v)[4]) {
    float t0 = acc[0] - mu * cs.x, t1 = acc[1] - mu * cs.y, t2 = acc[2] - mu * cs.z, t3 = acc[3] - mu * cs.w;
    if (!PACKED) asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
    v[0] = rs * t0 + bq.x; v[1] = rs * t1 + bq.y; v[2] = rs * t2 + bq.z; v[3] = rs * t3 + bq.w;
    if (!PACKED) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
}

// rows: 64 lanes = 16 pixel rows x 4 channel quads per fragment, NF fragments per wave and iteration (the product's 64 x 64 tile has 2 x 2 per wave)
template <bool PACKED, int NF>
__global__ __launch_bounds__(256) void k_repro(const float* __restrict__ acc_in, const double* __restrict__ stats, const float* __restrict__ colsum,
                                               const float* __restrict__ bias, const float* __restrict__ want, float inv_c, float eps, int iters, int noise,
                                               unsigned long long* bad, unsigned* first_bad, float* sink) {
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fk = lane >> 4;
    if (wave != 0 && noise) {
        // the other three waves of the workgroup: keep the SIMDs' other issue slots busy with different pipelines
        double d = 1.0 + lane * 1e-3; f4 c = {0.f, 0.f, 0.f, 0.f}; s8 a = {1, 2, 3, 4, 5, 6, 7, 8}; float x = lane;
        for (int it = 0; it < iters * NF; ++it) {
            if (wave == 1) { d = 1.0 / sqrt(d + 1.5); d = d * d + 0.25; }
            if (wave == 2) { c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c, 0, 0, 0); }
            if (wave == 3) { lds[(lane * 17 + it) & 4095] = x; x += lds[(lane * 5 + it * 3) & 4095] + acc_in[(it * 64 + lane) & 4095]; }
        }
        if (d == 123.0 || c[0] == 7.f || x == 3.f) sink[0] = 1.f;
        return;
    }
    if (wave != 0) return;
    unsigned long long nbad = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int row = (it * NF + f) * 16 + fr;                      // pixel row
            const double2 sq = *reinterpret_cast<const double2*>(stats + 2 * row);
            const double mean = sq.x * (double)inv_c, var = sq.y * (double)inv_c - mean * mean;
            const float mu = (float)mean, rs = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps));
            const float4 a4 = *reinterpret_cast<const float4*>(acc_in + (size_t)row * 16 + fk * 4);
            const float4 cs = *reinterpret_cast<const float4*>(colsum + fk * 4), bq = *reinterpret_cast<const float4*>(bias + fk * 4);
            const float acc[4] = {a4.x, a4.y, a4.z, a4.w};
            float v[4];
            ln_quad<PACKED>(acc, mu, rs, cs, bq, v);
            const float4 w = *reinterpret_cast<const float4*>(want + (size_t)row * 16 + fk * 4);
            const bool ok = __float_as_uint(v[0]) == __float_as_uint(w.x) && __float_as_uint(v[1]) == __float_as_uint(w.y) &&
                            __float_as_uint(v[2]) == __float_as_uint(w.z) && __float_as_uint(v[3]) == __float_as_uint(w.w);
            if (!ok) { if (!nbad) atomicMin(first_bad, (unsigned)row); ++nbad; }
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main(int argc, char** argv) {
    const int iters = 64, NF = 4, rows = iters * NF * 16, launches = argc > 1 ? atoi(argv[1]) : 200, blocks = 2048;
    std::vector<float> acc((size_t)rows * 16), cs(16), bq(16), want((size_t)rows * 16);
    std::vector<double> st((size_t)rows * 2);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    const float inv_c = 1.f / 320.f, eps = 1e-5f;
    for (auto& x : acc) x = 4.f * rnd();
    for (auto& x : cs) x = 3.f * rnd();
    for (auto& x : bq) x = rnd();
    for (int r = 0; r < rows; ++r) { const double m = rnd(), sd = 0.2 + fabs(rnd()); st[2 * r] = m * 320.0; st[2 * r + 1] = (sd * sd + m * m) * 320.0; }
    for (int r = 0; r < rows; ++r) {
        const double mean = st[2 * r] * (double)inv_c, var = st[2 * r + 1] * (double)inv_c - mean * mean;
        const float mu = (float)mean, rs = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps));
        for (int c = 0; c < 16; ++c) want[(size_t)r * 16 + c] = fmaf(rs, fmaf(-mu, cs[c], acc[(size_t)r * 16 + c]), bq[c]);
    }
    float *d_acc, *d_cs, *d_bq, *d_want, *d_sink; double* d_st; unsigned long long* d_bad; unsigned* d_first;
    CK(hipMalloc(&d_acc, acc.size() * 4)); CK(hipMalloc(&d_cs, 64)); CK(hipMalloc(&d_bq, 64)); CK(hipMalloc(&d_want, want.size() * 4)); CK(hipMalloc(&d_sink, 4));
    CK(hipMalloc(&d_st, st.size() * 8)); CK(hipMalloc(&d_bad, 8)); CK(hipMalloc(&d_first, 4));
    CK(hipMemcpy(d_acc, acc.data(), acc.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_cs, cs.data(), 64, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_bq, bq.data(), 64, hipMemcpyHostToDevice)); CK(hipMemcpy(d_want, want.data(), want.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_st, st.data(), st.size() * 8, hipMemcpyHostToDevice));
    for (int noise = 0; noise < 2; ++noise)
        for (int packed = 0; packed < 2; ++packed) {
            CK(hipMemset(d_bad, 0, 8)); CK(hipMemset(d_first, 0xff, 4));
            for (int l = 0; l < launches; ++l) {
                if (packed) hipLaunchKernelGGL((k_repro<true, NF>), dim3(blocks), dim3(256), 0, 0, d_acc, d_st, d_cs, d_bq, d_want, inv_c, eps, iters, noise, d_bad, d_first, d_sink);
                else hipLaunchKernelGGL((k_repro<false, NF>), dim3(blocks), dim3(256), 0, 0, d_acc, d_st, d_cs, d_bq, d_want, inv_c, eps, iters, noise, d_bad, d_first, d_sink);
            }
            CK(hipDeviceSynchronize());
            unsigned long long bad; unsigned first;
            CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost));
            printf("%s epilogue, %s: %llu wrong quads of %.3e (first wrong row %d)\n", packed ? "PACKED" : "scalar", noise ? "with three co-resident waves of other pipelines" : "alone",
                   bad, (double)launches * blocks * rows * 4, bad ? (int)first : -1);
        }
    return 0;
}
