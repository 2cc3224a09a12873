// Division of a non-negative 31-bit integer by a launch-time constant as a multiply and a shift (no integer divide instruction on gfx950: a 32-bit division is ~30
// vector instructions, a 64-bit one ~150).  Plain C++: compiled by hipcc into the kernels and by g++ into tests/test_fastdiv.py.
#pragma once
#include <cstdint>

// n / d for 0 <= n < 2^31 as (n * mul) >> (32 + shr): s = ceil(log2 d), mul = floor(2^(31 + s) / d) + 1, shr = s - 1 (mul * d = 2^(31 + s) + e with 0 < e <= d <= 2^s, so
// the error term n e / (d 2^(31 + s)) stays below 1 / d).  d == 1 is mul = 0: the quotient is n.
inline void mf_fastdiv(uint32_t d, uint32_t* mul, uint32_t* shr) {
    if (d <= 1) { *mul = 0; *shr = 0; return; }
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;
    *mul = (uint32_t)(((1ull << (31 + s)) / d) + 1);
    *shr = s - 1;
}
// the kernels' form on the host (tests, launch-side checks)
inline int mf_fdiv_host(int n, uint32_t mul, uint32_t shr) { return mul ? (int)((uint32_t)(((uint64_t)(uint32_t)n * mul) >> 32) >> shr) : n; }
#ifdef __HIPCC__
__device__ __forceinline__ int mf_fdiv(int n, uint32_t mul, uint32_t shr) { return mul ? (int)(__umulhi((uint32_t)n, mul) >> shr) : n; }
#endif
