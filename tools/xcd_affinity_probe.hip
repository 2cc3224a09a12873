// Does a line written by kernel A on XCD x hit in L2 when kernel B reads it on the SAME XCD (vs another XCD)?   (GPU box)
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_affinity_probe.hip -o /tmp/xcd_probe && /tmp/xcd_probe
// Block b runs on XCD b % 8 (observed, MI355X_MICROARCH.md).  Producer block b writes chunk b; consumer block b reads chunk (b + shift) % nblocks
// with a dependent-latency-exposing pattern (one 16-byte load per lane per iteration, a short chain) and a streaming pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_write(uint4* buf, int chunk_vec, unsigned seed) {
    uint4* p = buf + (size_t)blockIdx.x * chunk_vec;
    for (int i = threadIdx.x; i < chunk_vec; i += 256) p[i] = make_uint4(seed + i, blockIdx.x, i, seed);
}
__global__ __launch_bounds__(256) void k_read(const uint4* buf, int chunk_vec, int shift, unsigned* out, int xcd_check) {
    const int src = (blockIdx.x + shift) % gridDim.x;
    const uint4* p = buf + (size_t)src * chunk_vec;
    unsigned acc = 0;
    // 4 loads in flight per lane
    for (int i = threadIdx.x; i + 768 < chunk_vec; i += 1024) {
        const uint4 a = p[i], b = p[i + 256], c = p[i + 512], d = p[i + 768];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
    if (xcd_check && threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out[1 + blockIdx.x] = x & 0xf;
    }
}

int main() {
    const int nblocks = 256;
    for (int chunk_kb : {16, 64, 256}) {
        const int chunk_vec = chunk_kb * 1024 / 16;
        uint4* buf; unsigned* out;
        CK(hipMalloc(&buf, (size_t)nblocks * chunk_vec * 16));
        CK(hipMalloc(&out, 4 * (nblocks + 1)));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int shift : {0, 8, 1, 3, 0}) {
            float tot = 0;
            const int it = 20;
            for (int i = 0; i < it; ++i) {
                hipLaunchKernelGGL(k_write, dim3(nblocks), dim3(256), 0, 0, buf, chunk_vec, (unsigned)i);
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k_read, dim3(nblocks), dim3(256), 0, 0, buf, chunk_vec, shift, out, i == 0);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (i >= 5) tot += ms;
            }
            std::vector<unsigned> h(nblocks + 1);
            CK(hipMemcpy(h.data(), out, 4 * (nblocks + 1), hipMemcpyDeviceToHost));
            int ok = 0;
            for (int b = 0; b < nblocks; ++b) ok += (int)h[1 + b] == b % 8;
            printf("chunk %4d KB (total %5.1f MB) shift %d (consumer XCD %s producer's): read kernel %.2f us  -> %.2f TB/s   [blocks on XCD b%%8: %d/%d]\n",
                   chunk_kb, nblocks * chunk_kb / 1024.0, shift, shift % 8 == 0 ? "==" : "!=", tot / 15 * 1e3,
                   nblocks * chunk_kb * 1024.0 / (tot / 15 * 1e-3) / 1e12, ok, nblocks);
        }
        CK(hipFree(buf)); CK(hipFree(out));
    }
    return 0;
}
