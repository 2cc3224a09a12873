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
