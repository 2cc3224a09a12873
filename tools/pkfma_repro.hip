// Reproducer for the "wrong channel now and then" finding of round 5 -- a gfx950 hardware erratum, found in round 6:
//
//     a packed-fp32 VALU instruction whose LOW result takes the HIGH register of a 64-bit source (VOP3P op_sel bit set, e.g. `v_pk_fma_f32 ... op_sel:[0,1,0]`)
//     returns a wrong low half in lanes 48..63 when another wave of the same SIMD is issuing MFMAs.
//
// How it was found: in the LayerNorm-folding epilogue of k_conv_igemm the compiler kept (mean, rstd) in one register pair and, for the LAST fragment of a wave only,
// multiplied by rstd with `v_pk_fma_f32 vD, vT, v[mean:rstd], vB op_sel:[0,1,0]`; every wrong element of the failing layer was a low half of exactly those instructions,
// in lanes 48..63, and equal to the bias -- the product term read as zero (tools/pkfma_dump_analyze.py).  The other fragments used op_sel_hi:[1,0,1] (the HIGH result
// takes the LOW register) and never failed.  The workgroups of that kernel run their epilogues while neighbouring waves are still in their MFMA loops.
//
// This program: 512-thread workgroups; waves 0-3 run ONE packed instruction per iteration (inline asm, fixed registers v[100:105]) against the scalar instructions on the
// same operands and count mismatches per lane quarter and by kind of wrong value; waves 4-7 -- one per SIMD, beside a tester -- run nothing / MFMAs / LDS-DMA /
// LDS traffic / global traffic / fp64 FMAs.          hipcc --offload-arch=gfx950 -O3 tools/pkfma_repro.hip -o tools/bin/pkfma_repro      (build here, run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));

enum { N_NONE, N_MFMA, N_LDSDMA, N_LDS, N_GLOBAL, N_FP64 };
static const char* noise_names[] = {"alone", "beside MFMAs", "beside LDS-DMA", "beside LDS reads/writes", "beside global loads/stores", "beside fp64 FMAs"};

template <int NOISE>
__device__ __forceinline__ void noise_wave(const float* __restrict__ t_in, float* sink, int iters, float* lds) {
    const int lane = threadIdx.x & 63;
    f4v c = {0.f, 0.f, 0.f, 0.f}; s8v a = {1, 2, 3, 4, 5, 6, 7, 8}; double d = 1.0 + lane * 1e-3; float x = lane;
    for (int it = 0; it < iters * 6; ++it) {
        if (NOISE == N_MFMA) { c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c, 0, 0, 0); }
        if (NOISE == N_LDSDMA) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(t_in + ((it * 256 + lane * 4) & 65535)),
                                             (__attribute__((address_space(3))) void*)(lds + (threadIdx.x >> 6) * 256), 16, 0, 0);
            if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (NOISE == N_LDS) { lds[(lane * 17 + it) & 4095] = x; x += lds[(lane * 5 + it * 3) & 4095]; }
        if (NOISE == N_GLOBAL) { x += t_in[(it * 64 + lane * 33) & 65535]; if ((it & 3) == 0) sink[256 + ((blockIdx.x * 64 + lane) & 4095)] = x; }
        if (NOISE == N_FP64) { d = __builtin_fma(d, 0.999, 0.25); d = __builtin_fma(d, d, -0.1); }
    }
    if (NOISE == N_LDSDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (c[0] == 7.f || d == 123.0 || x == 3.f) sink[0] = 1.f;
}

// One case = one packed instruction with its modifiers, and the two scalar instructions that say what its halves must be.  Sources: A = v[100:101], B = v[102:103],
// C = v[104:105], filled from (a.x, a.y), (b.x, b.y), (c.x, c.y).
enum { C_FMA_PLAIN, C_FMA_HI_FROM_LO_1, C_FMA_LO_FROM_HI_1, C_FMA_LO_FROM_HI_0, C_FMA_LO_FROM_HI_2, C_FMA_BOTH_FROM_HI_1, C_MUL_LO_FROM_HI_1, C_ADD_LO_FROM_HI_1,
       C_FMA16_LO_FROM_HI_1, C_FMA_LO_FROM_HI_1_NOP, C_ADD_LO_FROM_HI_0, C_MOV_SEL10, C_MOV_SEL01, N_CASES };
static const char* case_names[] = {
    "v_pk_fma_f32                                   (no select)",
    "v_pk_fma_f32 op_sel_hi:[1,0,1]                 (high result <- src1 LOW)",
    "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]  (low result <- src1 HIGH, high <- src1 low: a swap)",
    "v_pk_fma_f32 op_sel:[1,0,0]                    (low result <- src0 HIGH)",
    "v_pk_fma_f32 op_sel:[0,0,1]                    (low result <- src2 HIGH)",
    "v_pk_fma_f32 op_sel:[0,1,0]                    (both results <- src1 HIGH: the kernel's form)",
    "v_pk_mul_f32 op_sel:[0,1]                      (both results <- src1 HIGH)",
    "v_pk_add_f32 op_sel:[0,1]                      (both results <- src1 HIGH)",
    "v_pk_fma_f16 op_sel:[0,1,0]                    (16-bit halves of ONE register: low result <- src1 high half)",
    "v_pk_fma_f32 op_sel:[0,1,0] after s_nop 7      (the kernel's form, 8 wait states behind its producers)",
    "v_pk_add_f32 op_sel:[1,0]                      (low result <- src0 HIGH)",
    "v_pk_mov_b32 op_sel:[1,0]                      (D = (src0 HIGH, src1 low))",
    "v_pk_mov_b32 op_sel:[0,1]                      (D = (src0 low, src1 HIGH))",
};

template <int CASE, int NOISE>
__global__ __launch_bounds__(512) void k_repro(const float* __restrict__ t_in, int iters, unsigned long long* bad, unsigned* hist, float* sink) {
    __shared__ float lds[4096];
    if (threadIdx.x >= 256) { noise_wave<NOISE>(t_in, sink, iters, lds); return; }
    const int lane = threadIdx.x & 63;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    unsigned long long nbad = 0;
    for (int it = 0; it < iters; ++it) {
        const int idx = (gid * 7 + it * 8191) & 32767;
        const f2v a = {t_in[idx], t_in[idx + 32768]}, b = {t_in[idx + 65536] * 0.25f + 1.5f, t_in[idx + 98304] * 0.25f - 1.5f}, c = {t_in[(idx + 11) & 32767], t_in[(idx + 77) & 32767]};
        f2v r; float w0, w1;
#define FILL "v_mov_b32 v100, %1\n\tv_mov_b32 v101, %2\n\tv_mov_b32 v102, %3\n\tv_mov_b32 v103, %4\n\tv_mov_b32 v104, %5\n\tv_mov_b32 v105, %6\n\t"
#define OPS : "=&v"(r) : "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y), "v"(c.x), "v"(c.y) : "v100", "v101", "v102", "v103", "v104", "v105"
        if (CASE == C_FMA_PLAIN)          { asm volatile(FILL "v_pk_fma_f32 %0, v[100:101], v[102:103], v[104:105]" OPS); w0 = __builtin_fmaf(a.x, b.x, c.x); w1 = __builtin_fmaf(a.y, b.y, c.y); }
        if (CASE == C_FMA_HI_FROM_LO_1)   { asm volatile(FILL "v_pk_fma_f32 %0, v[100:101], v[102:103], v[104:105] op_sel_hi:[1,0,1]" OPS); w0 = __builtin_fmaf(a.x, b.x, c.x); w1 = __builtin_fmaf(a.y, b.x, c.y); }
        if (CASE == C_FMA_LO_FROM_HI_1)   { asm volatile(FILL "v_pk_fma_f32 %0, v[100:101], v[102:103], v[104:105] op_sel:[0,1,0] op_sel_hi:[1,0,1]" OPS); w0 = __builtin_fmaf(a.x, b.y, c.x); w1 = __builtin_fmaf(a.y, b.x, c.y); }
        if (CASE == C_FMA_LO_FROM_HI_0)   { asm volatile(FILL "v_pk_fma_f32 %0, v[100:101], v[102:103], v[104:105] op_sel:[1,0,0]" OPS); w0 = __builtin_fmaf(a.y, b.x, c.x); w1 = __builtin_fmaf(a.y, b.y, c.y); }
        if (CASE == C_FMA_LO_FROM_HI_2)   { asm volatile(FILL "v_pk_fma_f32 %0, v[100:101], v[102:103], v[104:105] op_sel:[0,0,1]" OPS); w0 = __builtin_fmaf(a.x, b.x, c.y); w1 = __builtin_fmaf(a.y, b.y, c.y); }
        if (CASE == C_FMA_BOTH_FROM_HI_1) { asm volatile(FILL "v_pk_fma_f32 %0, v[100:101], v[102:103], v[104:105] op_sel:[0,1,0]" OPS); w0 = __builtin_fmaf(a.x, b.y, c.x); w1 = __builtin_fmaf(a.y, b.y, c.y); }
        if (CASE == C_FMA_LO_FROM_HI_1_NOP) { asm volatile(FILL "s_nop 7\n\tv_pk_fma_f32 %0, v[100:101], v[102:103], v[104:105] op_sel:[0,1,0]" OPS); w0 = __builtin_fmaf(a.x, b.y, c.x); w1 = __builtin_fmaf(a.y, b.y, c.y); }
        if (CASE == C_MUL_LO_FROM_HI_1)   { asm volatile(FILL "v_pk_mul_f32 %0, v[100:101], v[102:103] op_sel:[0,1]" OPS); w0 = a.x * b.y; w1 = a.y * b.y; }
        if (CASE == C_ADD_LO_FROM_HI_0)   { asm volatile(FILL "v_pk_add_f32 %0, v[100:101], v[102:103] op_sel:[1,0]" OPS); w0 = a.y + b.x; w1 = a.y + b.y; }
        if (CASE == C_MOV_SEL10)          { asm volatile(FILL "v_pk_mov_b32 %0, v[100:101], v[102:103] op_sel:[1,0]" OPS); w0 = a.y; w1 = b.x; }
        if (CASE == C_MOV_SEL01)          { asm volatile(FILL "v_pk_mov_b32 %0, v[100:101], v[102:103] op_sel:[0,1]" OPS); w0 = a.x; w1 = b.y; }
        if (CASE == C_ADD_LO_FROM_HI_1)   { asm volatile(FILL "v_pk_add_f32 %0, v[100:101], v[102:103] op_sel:[0,1]" OPS); w0 = a.x + b.y; w1 = a.y + b.y; }
        if (CASE == C_FMA16_LO_FROM_HI_1) {     // the 16-bit packed form (the selects pick halves of ONE 32-bit register), against the same instruction on a pre-swizzled src1
            typedef __fp16 h2v __attribute__((ext_vector_type(2)));
            const h2v ha = __builtin_amdgcn_cvt_pkrtz(a.x, a.y), hb = __builtin_amdgcn_cvt_pkrtz(b.x, b.y), hc = __builtin_amdgcn_cvt_pkrtz(c.x, c.y);
            unsigned x0, x1;
            asm volatile("v_mov_b32 v100, %2\n\tv_mov_b32 v102, %3\n\tv_mov_b32 v104, %4\n\tv_pk_fma_f16 %0, v100, v102, v104 op_sel:[0,1,0]\n\t"
                         "v_lshrrev_b32 v103, 16, v102\n\tv_and_b32 v101, 0xffff0000, v102\n\tv_or_b32 v103, v103, v101\n\tv_pk_fma_f16 %1, v100, v103, v104"
                         : "=&v"(x0), "=&v"(x1) : "v"(ha), "v"(hb), "v"(hc) : "v100", "v101", "v102", "v103", "v104");
            r.x = __uint_as_float(x0 & 0xffffu); r.y = __uint_as_float(x0 >> 16); w0 = __uint_as_float(x1 & 0xffffu); w1 = __uint_as_float(x1 >> 16);
        }
#undef FILL
#undef OPS
        asm volatile("" : "+v"(w0), "+v"(w1));
        const bool lo_bad = __float_as_uint(r.x) != __float_as_uint(w0), hi_bad = __float_as_uint(r.y) != __float_as_uint(w1);
        if (lo_bad || hi_bad) {
            ++nbad;
            // what IS the wrong low half?  the addend alone (the product read as zero: what the kernel showed), the unselected register used, zero, or something else
            int kind = 3;
            if (lo_bad) {
                if (r.x == c.x) kind = 0;
                else if (r.x == __builtin_fmaf(a.x, b.x, c.x) || r.x == a.x * b.x || r.x == a.x + b.x || r.x == a.x) kind = 1;
                else if (r.x == 0.f) kind = 2;
            }
            atomicAdd(hist + (lo_bad ? 0 : 4) + (lane >> 4), 1u);
            if (lo_bad) atomicAdd(hist + 8 + kind, 1u);
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

static float* d_t; static unsigned long long* d_bad; static unsigned* d_hist; static float* d_sink; static int launches = 10;

template <int CASE, int NOISE>
static void run() {
    const int iters = 256, blocks = 4096;
    CK(hipMemset(d_bad, 0, 8)); CK(hipMemset(d_hist, 0, 16 * 4));
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((k_repro<CASE, NOISE>), dim3(blocks), dim3(NOISE ? 512 : 256), 0, 0, d_t, iters, d_bad, d_hist, d_sink);
    CK(hipDeviceSynchronize());
    unsigned long long bad; unsigned h[16];
    CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h, d_hist, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-104s %-26s: %10llu wrong of %.2e", case_names[CASE], noise_names[NOISE], bad, (double)launches * blocks * 256 * iters);
    if (bad)
        printf("  | low half wrong in lanes 0-15/16-31/32-47/48-63: %u/%u/%u/%u; high half only: %u/%u/%u/%u | the wrong low half is: the addend alone %u, the unselected register's result %u, zero %u, other %u",
               h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11]);
    printf("\n");
}

int main(int argc, char** argv) {
    launches = argc > 1 ? atoi(argv[1]) : 10;
    std::vector<float> t(131072);
    srand(1);
    for (auto& x : t) x = (float)((double)rand() / RAND_MAX * 8.0 - 4.0);
    CK(hipMalloc(&d_t, t.size() * 4)); CK(hipMalloc(&d_bad, 8)); CK(hipMalloc(&d_hist, 16 * 4)); CK(hipMalloc(&d_sink, 8192 * 4));
    CK(hipMemcpy(d_t, t.data(), t.size() * 4, hipMemcpyHostToDevice));
    // every form alone, then beside MFMAs; the kernel's form beside every other pipeline
    run<C_FMA_PLAIN, N_NONE>(); run<C_FMA_HI_FROM_LO_1, N_NONE>(); run<C_FMA_BOTH_FROM_HI_1, N_NONE>(); run<C_FMA_LO_FROM_HI_1, N_NONE>();
    run<C_FMA_PLAIN, N_MFMA>(); run<C_FMA_HI_FROM_LO_1, N_MFMA>();
    run<C_FMA_BOTH_FROM_HI_1, N_MFMA>(); run<C_FMA_LO_FROM_HI_1, N_MFMA>(); run<C_FMA_LO_FROM_HI_0, N_MFMA>(); run<C_FMA_LO_FROM_HI_2, N_MFMA>();
    run<C_MUL_LO_FROM_HI_1, N_MFMA>(); run<C_ADD_LO_FROM_HI_1, N_MFMA>(); run<C_ADD_LO_FROM_HI_0, N_MFMA>(); run<C_MOV_SEL10, N_MFMA>(); run<C_MOV_SEL01, N_MFMA>(); run<C_FMA16_LO_FROM_HI_1, N_MFMA>(); run<C_FMA_LO_FROM_HI_1_NOP, N_MFMA>();
    run<C_FMA_BOTH_FROM_HI_1, N_LDSDMA>(); run<C_FMA_BOTH_FROM_HI_1, N_LDS>(); run<C_FMA_BOTH_FROM_HI_1, N_GLOBAL>(); run<C_FMA_BOTH_FROM_HI_1, N_FP64>();
    return 0;
}
