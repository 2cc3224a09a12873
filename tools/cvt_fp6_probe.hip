// What v_cvt_scalef32_pk32_fp6_f16 / v_cvt_scalef32_2xpk16_fp6_f32 (gfx950) compute: element order, what the scale operand does, rounding of ties,
// saturation.  hipcc --offload-arch=gfx950 -O2 tools/cvt_fp6_probe.hip -o /tmp/cvt_probe && /tmp/cvt_probe     (GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v32h __attribute__((ext_vector_type(32)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
__global__ void k(const float* x, unsigned* out, float scale) {
    v16f a, b;
    v32h h;
    for (int i = 0; i < 16; ++i) { a[i] = x[i]; b[i] = x[16 + i]; }
    for (int i = 0; i < 32; ++i) h[i] = (_Float16)x[i];
    v6u r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
    v6u r2 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(h, scale);
    for (int i = 0; i < 6; ++i) { out[i] = r[i]; out[6 + i] = r2[i]; }
}
static float dec(unsigned c) {
    const int s = c >> 5, e = (c >> 3) & 3, m = c & 7;
    const float v = e == 0 ? m * 0.125f : ldexpf(1.f + m * 0.125f, e - 1);
    return s ? -v : v;
}
static unsigned code(const unsigned* w, int t) {
    const int bit = 6 * t;
    unsigned long long v = w[bit >> 5];
    if ((bit >> 5) + 1 < 6) v |= (unsigned long long)w[(bit >> 5) + 1] << 32;
    return (unsigned)(v >> (bit & 31)) & 63u;
}
int main() {
    const float vals[32] = {0.f, 0.0625f, 0.125f, 0.1875f, 0.3125f, 0.4375f, 1.0625f, 1.1875f, 2.125f, 2.375f, 4.25f, 4.75f, 7.5f, 7.75f, 8.f, 100.f,
                            -0.0625f, -0.1875f, -0.3125f, -1.0625f, -1.1875f, -2.125f, -2.375f, -4.25f, -4.75f, -7.75f, -100.f, 0.06f, 0.07f, 3.f, 5.f, 6.9f};
    float* dx; unsigned* dout;
    hipMalloc(&dx, 32 * 4); hipMalloc(&dout, 12 * 4);
    for (float scale : {1.f, 2.f, 0.5f, 3.f, 0.25f}) {
        for (float mul : {1.f}) {
            std::vector<float> x(32);
            for (int i = 0; i < 32; ++i) x[i] = vals[i] * mul * scale;        // if the instruction divides by the scale these land on the table above
            hipMemcpy(dx, x.data(), 128, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, dx, dout, scale);
            unsigned o[12];
            hipMemcpy(o, dout, 48, hipMemcpyDeviceToHost);
            printf("scale %.2f (inputs = table * scale)\n   input      f32->fp6   f16->fp6\n", scale);
            for (int i = 0; i < 32; ++i) printf("  %9.4f  %9.4f  %9.4f\n", x[i], dec(code(o, i)), dec(code(o + 6, i)));
        }
    }
    return 0;
}
