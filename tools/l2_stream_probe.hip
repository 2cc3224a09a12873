// How many operand bytes per clock a CU can pull from L2, by landing zone: LDS (global_load_lds DMA, double-buffered stages as in
// k_conv_igemm), VGPRs (global_load_dwordx4 into a register ring), or both at once.  Tests the Little's-law reading of the ~17 B/clk/CU
// L2 -> LDS figure (profiles/r01_igemm_bandwidth_study.md): if registers add landing capacity, "both" exceeds "LDS only".
//   hipcc --offload-arch=gfx950 -O3 tools/l2_stream_probe.hip -o gpurun_out/l2_stream_probe   (build here, run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// STAGE_KB per LDS stage (2 stages), RV 16-byte register loads per lane and iteration (ring of 2 iterations).
// Each workgroup walks its own slice of a `span`-byte window (span <= L2/MALL size -> cache-resident like re-read weights).
// ROWB > 0: the DMA gathers ROWB-byte row segments `stride` bytes apart (an im2col row of BK channels), 1024 / ROWB rows per instruction
template <int STAGE_KB, int RV, bool USE_LDS, bool USE_REG, int ROWB = 0>
__global__ __launch_bounds__(256) void k_stream(const char* __restrict__ src, size_t span, int iters, unsigned* sink, int stride = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int STAGE = STAGE_KB * 1024;
    constexpr int CH = STAGE / 1024 / 4;                        // 1-KiB DMA chunks per wave per stage
    const size_t wg_bytes = ROWB ? (size_t)(STAGE / ROWB) * stride : (size_t)STAGE + (size_t)RV * 256 * 16;
    size_t off = ((size_t)blockIdx.x * 7919 * wg_bytes) % span;
    u32x4 ring0[RV > 0 ? RV : 1], ring1[RV > 0 ? RV : 1];       // two named rings: a runtime stage index would push them to scratch
    unsigned acc = 0;
    auto issue = [&](int it, int st, u32x4* ring) __attribute__((always_inline)) {
        const char* base = src + (off + (size_t)it * wg_bytes * gridDim.x) % span;
        if (USE_LDS) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const char* g = base + (size_t)(wave * CH + c) * 1024 + lane * 16;
                if (ROWB) { constexpr int KG = ROWB / 16; g = base + (size_t)((wave * CH + c) * (1024 / ROWB) + lane / KG) * stride + (lane % KG) * 16; }
                glds16(g, smem + st * STAGE + (wave * CH + c) * 1024);
            }
        }
        if (USE_REG) {
#pragma unroll
            for (int r = 0; r < RV; ++r) ring[r] = *reinterpret_cast<const u32x4*>(base + STAGE + ((size_t)r * 256 + tid) * 16);
        }
    };
    auto consume = [&](int st, const u32x4* ring) __attribute__((always_inline)) {
        wait_vm<(USE_LDS ? CH : 0) + (USE_REG ? RV : 0)>();       // the older of the two iterations in flight has landed
        __syncthreads();
        if (USE_LDS) acc ^= *reinterpret_cast<const unsigned*>(smem + st * STAGE + tid * 4);
        if (USE_REG) {
#pragma unroll
            for (int r = 0; r < RV; ++r) acc ^= ring[r].x ^ ring[r].w;
        }
        __syncthreads();
    };
    issue(0, 0, ring0);
    for (int it = 0; it + 2 < iters; it += 2) {
        issue(it + 1, 1, ring1);
        consume(0, ring0);
        issue(it + 2, 0, ring0);
        consume(1, ring1);
    }
    wait_vm<0>();
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int STAGE_KB, int RV, bool L, bool R, int ROWB = 0>
double run(const char* src, size_t span, int wgs, int iters, unsigned* sink, const char* what, int stride = 0) {
    auto kern = k_stream<STAGE_KB, RV, L, R, ROWB>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t lds = L ? 2 * STAGE_KB * 1024 : 1024;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds, 0, src, span, iters, sink, stride);
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds, 0, src, span, iters, sink, stride);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); ms /= 5;
    const double bytes = (double)wgs * iters * ((L ? STAGE_KB * 1024.0 : 0) + (R ? RV * 4096.0 : 0));
    const double tbs = bytes / (ms * 1e-3) / 1e12;
    printf("%-44s wgs %4d  %8.1f us  %6.2f TB/s  = %5.1f B/clk/CU at 2.4 GHz x 256 CUs\n", what, wgs, ms * 1e3, tbs, tbs * 1e12 / 256 / 2.4e9);
    return tbs;
}

int main(int argc, char** argv) {
    const size_t span = (size_t)(argc > 1 ? atoi(argv[1]) : 24) << 20;       // MiB window: 24 = L2-resident, 200 = MALL, 2048 = HBM
    const int iters = 400;
    char* src; unsigned* sink;
    hipMalloc(&src, span + (8 << 20)); hipMemset(src, 1, span + (8 << 20)); hipMalloc(&sink, 64);
    printf("window %zu MiB\n", span >> 20);
    for (int wgs : {512}) {
        run<32, 0, true, false>(src, span, wgs, iters, sink, "LDS only, 2 x 32 KB stages");
        run<64, 0, true, false>(src, span, wgs, iters, sink, "LDS only, 2 x 64 KB stages");
        run<0, 8, false, true>(src, span, wgs, iters, sink, "VGPR only, 2 x 8 x 16 B per lane (32 KB/WG)");
        run<0, 16, false, true>(src, span, wgs, iters, sink, "VGPR only, 2 x 16 x 16 B per lane (64 KB/WG)");
        run<32, 8, true, true>(src, span, wgs, iters, sink, "both: 32 KB LDS stage + 32 KB VGPR");
        run<32, 16, true, true>(src, span, wgs, iters, sink, "both: 32 KB LDS stage + 64 KB VGPR");
        run<16, 16, true, true>(src, span, wgs, iters, sink, "both: 16 KB LDS stage + 64 KB VGPR");
        for (int stride : {128, 256, 512, 640, 1024, 1280, 2560}) {
            char nm[96];
            snprintf(nm, sizeof(nm), "LDS 32 KB stage, 64 B rows, stride %d", stride);
            run<32, 0, true, false, 64>(src, span, wgs, iters, sink, nm, stride);
            snprintf(nm, sizeof(nm), "LDS 32 KB stage, 128 B rows, stride %d", stride);
            run<32, 0, true, false, 128>(src, span, wgs, iters, sink, nm, stride);
        }
    }
    return 0;
}
