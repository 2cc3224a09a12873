// Stand-alone timing of the GroupNorm-apply + SiLU + f16/FP6 conversion pass (mf_affine_silu_to_act_q, mf_aux.hip) at the VAE decoder's and the
// UNet's shapes: achieved HBM GB/s per shape for each kernel variant (MF_AFFQ_VARIANT=0: one thread = consecutive (pixel, block) entries with
// per-thread parameter loads; 1: the block uniform over a wave, parameters in SGPRs), and a bit-for-bit comparison of the variants' outputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=262144 tools/affine_q_probe.hip -o build_ab/affine_q_probe
//   (build here, run on the GPU box:  MF_AFFQ_VARIANT=0 build_ab/affine_q_probe ; MF_AFFQ_VARIANT=1 build_ab/affine_q_probe)
#include "../mere-fusion_amd/csrc/mf_aux.hip"
#include <cstdarg>
#include <cstring>
#include <vector>

void mf_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static uint64_t fnv(const void* p, size_t n) {
    const uint64_t* q = (const uint64_t*)p; uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n / 8; ++i) { h ^= q[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv) {
    struct Shape { int B, H, W, C; };
    std::vector<Shape> shapes = {{8, 256, 256, 128}, {8, 256, 256, 256}, {8, 128, 128, 256}, {8, 128, 128, 512}, {8, 64, 64, 512}, {8, 32, 32, 512},
                                 {64, 256, 256, 128}, {64, 128, 128, 256}, {64, 64, 64, 512}, {64, 32, 32, 320}, {64, 16, 16, 640}};
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    const bool sums = argc > 2 && atoi(argv[2]);
    for (const Shape& sh : shapes) {
        ActBuf x, y;
        x.C = y.C = sh.C; x.H = y.H = sh.H; x.W = y.W = sh.W; x.halo = 0; y.halo = 1;
        const size_t nx = (size_t)sh.B * x.per_batch(), ny = (size_t)sh.B * y.per_batch();
        CK(hipMalloc(&x.hi, nx * 2)); CK(hipMalloc(&x.lo, nx * 2)); CK(hipMalloc(&y.hi, ny * 2)); CK(hipMalloc(&y.lo, ny * 2));
        CK(hipMemset(y.hi, 0, ny * 2)); CK(hipMemset(y.lo, 0, ny * 2));
        {   // activations ~ N(0, 1) as (hi, lo) bf16 pairs, from a cheap hash
            std::vector<bf16_t> h(nx), l(nx);
            uint32_t st = 12345u;
            for (size_t i = 0; i < nx; ++i) {
                float a = 0.f;
                for (int k = 0; k < 4; ++k) { st = st * 1664525u + 1013904223u; a += (float)(st >> 8) * (1.f / 16777216.f) - 0.5f; }
                const float v = a * 1.7320508f;
                h[i] = mf_f2bf(v); l[i] = mf_f2bf(v - mf_bf2f(h[i]));
            }
            CK(hipMemcpy(x.hi, h.data(), nx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(x.lo, l.data(), nx * 2, hipMemcpyHostToDevice));
        }
        std::vector<float> sc((size_t)sh.B * sh.C), sf((size_t)sh.B * sh.C), po(sh.C);
        for (size_t i = 0; i < sc.size(); ++i) { sc[i] = 0.5f + (float)(i % 7) * 0.25f; sf[i] = -0.3f + (float)(i % 5) * 0.2f; }
        for (int c = 0; c < sh.C; ++c) po[c] = ldexpf(1.f, (c % 5) - 2);
        float *dsc, *dsf, *dpo;
        CK(hipMalloc(&dsc, sc.size() * 4)); CK(hipMalloc(&dsf, sf.size() * 4)); CK(hipMalloc(&dpo, po.size() * 4));
        CK(hipMemcpy(dsc, sc.data(), sc.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsf, sf.data(), sf.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dpo, po.data(), po.size() * 4, hipMemcpyHostToDevice));
        ActView xv; xv.buf = &x; xv.coff = 0; xv.C = sh.C;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int w = 0; w < 3; ++w) if (mf_affine_silu_to_act_q(xv, dsc, dsf, 1, y, sh.B, 0, dpo) != MF_OK) return 1;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) mf_affine_silu_to_act_q(xv, dsc, dsf, 1, y, sh.B, 0, dpo);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1000.0 / iters, bytes = (double)sh.B * sh.H * sh.W * sh.C * 8.0;
        printf("B %2d %3dx%-3d C %3d  %8.1f us  %6.0f GB/s", sh.B, sh.H, sh.W, sh.C, us, bytes / us * 1e-3);
        if (sums) {
            std::vector<char> oh(ny * 2), ol(ny * 2);
            CK(hipMemcpy(oh.data(), y.hi, ny * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(ol.data(), y.lo, ny * 2, hipMemcpyDeviceToHost));
            printf("  sum %016llx %016llx", (unsigned long long)fnv(oh.data(), ny * 2), (unsigned long long)fnv(ol.data(), ny * 2));
            // the other three instantiations (SiLU without post, post without SiLU, neither)
            for (int combo = 0; combo < 3; ++combo) {
                mf_affine_silu_to_act_q(xv, dsc, dsf, combo == 0, y, sh.B, 0, combo == 1 ? dpo : nullptr); CK(hipDeviceSynchronize());
                CK(hipMemcpy(oh.data(), y.hi, ny * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(ol.data(), y.lo, ny * 2, hipMemcpyDeviceToHost));
                printf(" | %016llx %016llx", (unsigned long long)fnv(oh.data(), ny * 2), (unsigned long long)fnv(ol.data(), ny * 2));
            }
        }
        printf("\n");
        hipFree(x.hi); hipFree(x.lo); hipFree(y.hi); hipFree(y.lo); hipFree(dsc); hipFree(dsf); hipFree(dpo);
    }
    return 0;
}
