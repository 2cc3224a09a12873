// Hash / tiled grid encoder helpers shared by the compat kernel (mf_nerf.hip) and the fused torso kernel (mf_nerf_torso.hip).
// Reference: ernerf/gridencoder/src/gridencoder.cu:35-72 (fast_hash / get_grid_index), :122-124 (per-level scale / resolution).
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

constexpr int GRID_MAX_L = 32;
struct GridLevels {                      // per-level constants, computed on the host exactly as gridencoder.cu:122-124 does
    float scale[GRID_MAX_L];
    uint32_t resolution[GRID_MAX_L];
    uint32_t offset[GRID_MAX_L];
    uint32_t hashmap_size[GRID_MAX_L];
};

// fast_hash / get_grid_index, gridencoder.cu:35-72
template <uint32_t D>
__device__ __forceinline__ uint32_t grid_index(uint32_t C, uint32_t gridtype, bool align_corners, uint32_t hashmap_size,
                                               uint32_t resolution, const uint32_t (&pos_grid)[D]) {
    constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t stride = 1, index = 0;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        if (stride <= hashmap_size) {
            index += pos_grid[d] * stride;
            stride *= align_corners ? resolution : (resolution + 1);
        }
    }
    if (gridtype == 0 && stride > hashmap_size) {
        index = 0;
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) index ^= pos_grid[d] * primes[d];
    }
    return (index % hashmap_size) * C;
}
