// Shared host-side helpers for libmerefusion_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <cstdlib>
#include <string>
#include "../../include/merefusion.h"

typedef uint16_t bf16_t;  // raw bf16 bits; kernels reinterpret as needed

void mf_set_error(const char* fmt, ...);

// MF_DEBUG: comma-separated words for development output -- "times" (per-workgroup s_memtime stamps of the conv kernels), "tune" (one line per measured layer in
// mf_*_tune).  One variable instead of one per facility (switch creep).
static inline bool mf_debug_has(const char* word) {
    const char* e = getenv("MF_DEBUG");
    if (!e) return false;
    const std::string s = std::string(",") + e + ",";
    return s.find(std::string(",") + word + ",") != std::string::npos;
}

#define MF_HIP(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            mf_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,  \
                         __LINE__);                                                        \
            return MF_ERR_HIP;                                                             \
        }                                                                                  \
    } while (0)

#define MF_REQUIRE(cond, ...)                                                              \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            mf_set_error(__VA_ARGS__);                                                     \
            return MF_ERR_INVALID;                                                         \
        }                                                                                  \
    } while (0)

// round-to-nearest-even fp32 -> bf16 (host)
static inline bf16_t mf_f2bf(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
static inline float mf_bf2f(bf16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

// gfx950 erratum guard.  A packed-fp32 VALU instruction whose op_sel selects the HIGH register of src1 for the LOW result (`v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32 ... op_sel:[0,1,..]`) returns a wrong low half in lanes 48..63 -- src1 read as zero -- when another wave of the same SIMD is issuing MFMAs
// (tools/pkfma_repro.hip: 0.09 % of executions on MI355X; it was round 5's "wrong channel now and then").  The compiler forms that instruction by
// itself when SLP-vectorised scalar code combines the two halves of one register pair: a product or sum of neighbours (`x * y`, `x + y` with (x, y) in one pair), a
// multiplier kept in the high half of a pair.  Passing one operand through mf_opaque() makes it a scalar of unknown origin at that point, so the two halves are no
// longer one vector to the optimiser.  tools/isa_scan.py (run by build.py after every link, and by the CPU test suite) fails if the form is in the library anywhere.
#if defined(__HIPCC__)
__device__ __forceinline__ float mf_opaque(float x) {
    asm volatile("" : "+v"(x));
    return x;
}
// (a + b) + (c + d) with every partial sum a scalar of its own (a pair of partial sums added to each other is the same hazard one level up)
// GroupNorm as a per-(sample, channel) affine: from the (sum, sum of squares) of a group, y = x * sc + sh with sc = rstd * gamma, sh = beta - mean * sc.
// One definition for k_gn_affine (mf_nn.hip) and the conversion kernel that forms the pair itself (mf_aux.hip): every product and the fused steps are spelled out,
// so both give the same bits whatever the contraction setting of the including file.
__device__ __forceinline__ void mf_gn_affine_pair(double sum, double sumsq, double inv_n, float eps, float gamma, float beta, float& sc, float& sh) {
    const double mean = sum * inv_n;
    const double var = fmax(__builtin_fma(-mean, mean, sumsq * inv_n), 0.0);
    const float rstd = rsqrtf((float)var + eps);
    sc = rstd * gamma;
    sh = __builtin_fmaf(-(float)mean, sc, beta);
}
__device__ __forceinline__ float mf_sum4(float a, float b, float c, float d) {
    const float ab = a + mf_opaque(b), cd = c + mf_opaque(d);
    return mf_opaque(ab) + mf_opaque(cd);
}
#endif
