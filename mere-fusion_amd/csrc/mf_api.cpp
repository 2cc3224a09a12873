// Library-level entry points of libmerefusion_hip.so: device selection and error reporting.
#include "mf_common.h"
#include <cstring>

static thread_local char g_err[1024] = "";

void mf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mf_last_error(void) { return g_err; }

extern "C" int mf_abi_version(void) { return 1; }

extern "C" int mf_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        mf_set_error("mf_init: no HIP device visible (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return MF_ERR_NODEVICE;
    }
    MF_REQUIRE(device >= 0 && device < n, "mf_init: device %d out of range (%d visible)", device, n);
    MF_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    MF_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        mf_set_error("mf_init: device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return MF_ERR_NODEVICE;
    }
    return MF_OK;
}
