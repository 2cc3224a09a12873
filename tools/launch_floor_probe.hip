// What does a dependent kernel boundary cost inside a replayed hipGraph on MI355X, and what does a device-wide barrier inside ONE persistent
// kernel cost?  Decides whether the launch-bound chains (Wav2Lip's ~60 small layers, the UNet's 367 ops) are worth running as a persistent
// multi-layer kernel.   hipcc --offload-arch=gfx950 -O3 tools/launch_floor_probe.hip -o tools/bin/launch_floor_probe  (build here, run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_tiny(float* p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}

// sense-reversing barrier on one counter: arrivals add 1; the last one of a generation flips `gen`
__device__ __forceinline__ void grid_barrier(unsigned* cnt, volatile unsigned* gen, unsigned nblocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = *gen;
        __threadfence();
        if (atomicAdd(cnt, 1u) == nblocks - 1) {
            *cnt = 0;
            __threadfence();
            atomicAdd((unsigned*)gen, 1u);
        } else {
            while (*gen == g) __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_persistent(float* p, int n, int rounds, unsigned* cnt, unsigned* gen) {
    for (int r = 0; r < rounds; ++r) {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = p[i] * 1.0001f + 1.f;
        grid_barrier(cnt, gen, gridDim.x);
    }
}

int main() {
    const int rounds = 200;
    hipStream_t s; CK(hipStreamCreate(&s));
    unsigned *cnt, *gen; CK(hipMalloc(&cnt, 8)); CK(hipMemset(cnt, 0, 8)); gen = cnt + 1;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int n : {256, 65536, 1 << 20}) {
        float* p; CK(hipMalloc(&p, (size_t)n * 4)); CK(hipMemset(p, 0, (size_t)n * 4));
        const int blocks = (n + 255) / 256;
        // (a) eager launches
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(256), 0, s, p, n);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < rounds; ++i) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(256), 0, s, p, n);
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("n=%8d blocks=%5d  eager: %.2f us per dependent kernel\n", n, blocks, ms * 1e3 / rounds);
        // (b) the same chain captured into a graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < rounds; ++i) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(256), 0, s, p, n);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("n=%8d blocks=%5d  graph: %.2f us per dependent kernel\n", n, blocks, ms * 1e3 / (5 * rounds));
        // (c) one persistent kernel, `rounds` grid barriers; 256 / 512 / 1024 resident workgroups
        for (int pb : {256, 512, 1024}) {
            hipLaunchKernelGGL(k_persistent, dim3(pb), dim3(256), 0, s, p, n, 3, cnt, gen);
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(k_persistent, dim3(pb), dim3(256), 0, s, p, n, rounds, cnt, gen);
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("n=%8d persistent %4d WGs: %.2f us per round (work + grid barrier)\n", n, pb, ms * 1e3 / rounds);
        }
        CK(hipFree(p));
    }
    return 0;
}
