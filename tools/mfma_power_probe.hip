// Energy per MAC of the MFMA shapes the f16 + FP6 convolution could be built from, at the board's power cap: a register-blocked MFMA loop over random
// operands (NA A-fragments x NB B-fragments per k-step, as a conv wave's FM x FN block; fresh fragments each step from a small register pool so the
// operand buses toggle), run for a few seconds per shape on every CU while a host thread samples hwmon (package power, sclk).  Prints TFLOP/s, W, MHz and
// pJ per MAC.  Shapes: f16 16x16x32 vs 32x32x16; block-scaled FP6 16x16x128 vs 32x32x64; and the convolution's mix (two f16 + one FP6 per pair).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_power_probe.hip -o build_ab/mfma_power_probe -lpthread     (build here, run on the GPU box)
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cctype>
#include <dirent.h>
#include <string>
#include <thread>
#include <vector>
#include <chrono>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// MODE 0: f16 16x16x32, 8 x 4 tiles (128 accumulator registers: the SP consumer's block).  MODE 1: f16 32x32x16, 4 x 2 tiles (128 registers).
// MODE 2: FP6 16x16x128 scaled, 8 x 4.  MODE 3: FP6 32x32x64 scaled, 4 x 2.
template <int MODE>
__global__ __launch_bounds__(512) void k_loop(const i32x4* __restrict__ src, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    // a pool of random operand fragments per lane: 12 f16 fragments (16 bytes), or 8 FP6 operands (8 registers, 6 of codes)
    constexpr bool FP6 = MODE >= 2;
    i32x4 pool[FP6 ? 1 : 12];
    i32x8 pool8[FP6 ? 8 : 1];
    if constexpr (!FP6) {
#pragma unroll
        for (int i = 0; i < 12; ++i) pool[i] = src[i * 64 + lane];
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const i32x4 lo = src[2 * i * 64 + lane], hi = src[(2 * i + 1) * 64 + lane];
            pool8[i] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], 0, 0};
        }
    }
    float s = 0.f;
    if constexpr (MODE == 0 || MODE == 2) {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int step = 0; step < 4; ++step) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (MODE == 0) {
                            const f16x8 a = __builtin_bit_cast(f16x8, pool[(i + step * 3) % 12]), b = __builtin_bit_cast(f16x8, pool[(8 + j + step * 5) % 12]);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i][j], 0, 0, 0);
                        } else {
                            const i32x8 a = pool8[(i + step * 3) & 7], b = pool8[(4 + j + step * 5) & 7];
                            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i][j], 2, 2, 0, 127, 0, 127);
                        }
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    } else if constexpr (MODE == 1 || MODE == 3) {
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int step = 0; step < 8; ++step) {       // 32x32x16: half the K per instruction -> twice the k-steps for the same K
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr (MODE == 1) {
                            const f16x8 a = __builtin_bit_cast(f16x8, pool[(i + step * 3) % 12]), b = __builtin_bit_cast(f16x8, pool[(8 + j + step * 5) % 12]);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i][j], 0, 0, 0);
                        } else {
                            const i32x8 a = pool8[(i + step * 3) & 7], b = pool8[(4 + j + step * 5) & 7];
                            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[i][j], 2, 2, 0, 127, 0, 127);
                        }
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

static std::string find_hwmon() {
    // the hwmon directory of THIS process's GPU (a box may expose its neighbours' too): /sys/bus/pci/devices/<bdf>/hwmon/hwmon*
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, 0) != hipSuccess) return "";
    for (char* c = bdf; *c; ++c) *c = (char)tolower(*c);
    const std::string root = std::string("/sys/bus/pci/devices/") + bdf + "/hwmon";
    DIR* d = opendir(root.c_str());
    if (!d) return "";
    std::string best;
    while (dirent* e = readdir(d)) {
        if (strncmp(e->d_name, "hwmon", 5) != 0) continue;
        best = root + "/" + e->d_name;
        break;
    }
    closedir(d);
    return best;
}
static double read_num(const std::string& path) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return -1;
    double v = -1;
    if (fscanf(f, "%lf", &v) != 1) v = -1;
    fclose(f);
    return v;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
static int run(const char* label, double macs_per_wave_iter, const i32x4* src, float* out, const std::string& hw, double seconds) {
    // calibrate iterations for ~`seconds` of one launch chain
    const int grid = 256;                        // one workgroup of 8 waves per CU: 2 waves per SIMD
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int iters = 2000;
    hipLaunchKernelGGL(k_loop<MODE>, dim3(grid), dim3(512), 0, 0, src, out, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_loop<MODE>, dim3(grid), dim3(512), 0, 0, src, out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const int launches = (int)(seconds * 1000.0 / ms) + 1;
    std::atomic<bool> stop{false};
    std::vector<double> pw, mhz;
    std::thread sampler([&] {
        const std::string pf = read_num(hw + "/power1_input") >= 0 ? hw + "/power1_input" : hw + "/power1_average";
        while (!stop.load()) {
            pw.push_back(read_num(pf) * 1e-6); mhz.push_back(read_num(hw + "/freq1_input") * 1e-6);
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    });
    CK(hipEventRecord(e0));
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(k_loop<MODE>, dim3(grid), dim3(512), 0, 0, src, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    stop.store(true); sampler.join();
    CK(hipEventElapsedTime(&ms, e0, e1));
    // the second half of the samples: the power controller has settled
    double p = 0, f = 0; int n = 0;
    for (size_t i = pw.size() / 2; i < pw.size(); ++i) { p += pw[i]; f += mhz[i]; ++n; }
    p /= n ? n : 1; f /= n ? n : 1;
    const double macs = macs_per_wave_iter * iters * (double)launches * grid * 8;
    const double tmacs = macs / (ms * 1e-3) * 1e-12;
    printf("%-34s %7.1f TFLOP/s  %6.0f W  %5.0f MHz  %5.3f pJ/MAC  (%d samples)\n", label, 2 * tmacs, p, f, p / (tmacs * 1e12) * 1e12, n);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    const std::string hw = find_hwmon();
    if (hw.empty()) { fprintf(stderr, "no amdgpu hwmon with a power reading\n"); return 1; }
    printf("hwmon %s  cap %.0f W\n", hw.c_str(), read_num(hw + "/power1_cap") * 1e-6);
    // operand pools: f16 N(0,1) values / random FP6 codes
    std::vector<uint16_t> hf(16 * 64 * 8);
    std::vector<uint32_t> hq(16 * 64 * 4);
    uint32_t st = 777u;
    for (auto& v : hf) {
        float a = 0.f;
        for (int k = 0; k < 4; ++k) { st = st * 1664525u + 1013904223u; a += (float)(st >> 8) * (1.f / 16777216.f) - 0.5f; }
        const _Float16 h = (_Float16)(a * 1.7320508f);
        memcpy(&v, &h, 2);
    }
    for (auto& v : hq) { st = st * 1664525u + 1013904223u; v = st ^ (st >> 13); }
    i32x4 *df, *dq; float* out;
    CK(hipMalloc(&df, hf.size() * 2)); CK(hipMalloc(&dq, hq.size() * 4)); CK(hipMalloc(&out, 512 * 256 * 4));
    CK(hipMemcpy(df, hf.data(), hf.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    // MACs per wave per loop iteration
    if (run<0>("f16 16x16x32, 8x4 tiles", 4.0 * 32 * 16 * 16 * 32, df, out, hw, seconds)) return 1;
    if (run<1>("f16 32x32x16, 4x2 tiles", 8.0 * 8 * 32 * 32 * 16, df, out, hw, seconds)) return 1;
    if (run<2>("FP6 16x16x128 scaled, 8x4 tiles", 4.0 * 32 * 16 * 16 * 128, dq, out, hw, seconds)) return 1;
    if (run<3>("FP6 32x32x64 scaled, 4x2 tiles", 8.0 * 8 * 32 * 32 * 64, dq, out, hw, seconds)) return 1;
    return 0;
}
