// Prints what ds_read_b64_tr_b16 returns for a lane-linear 64 x 4 image (element value = its LDS index), to pin the
// transpose semantics mf_attn.hip relies on.  hipcc --offload-arch=gfx950 tools/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(short* dst) {
    __shared__ __attribute__((aligned(16))) short lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + l * 4));
    for (int j = 0; j < 4; ++j) dst[l * 4 + j] = v[j];
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            printf(" %3d", h[l * 4 + j]);
            ok &= h[l * 4 + j] == (l & 15) + j * 16 + (l >> 4) * 64;
        }
        printf("\n");
    }
    printf("matches lds[(l&15) + j*16 + (l>>4)*64]: %s\n", ok ? "yes" : "NO");
    return ok ? 0 : 1;
}
