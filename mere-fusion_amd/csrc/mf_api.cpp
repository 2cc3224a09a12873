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

extern "C" int mf_abi_version(void) { return 4; }

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

// ---- frame transport: host-side plumbing of the shared-memory frame ring (mere-fusion_amd/transport.py) -------------
// The reference pickles every generated frame through `mp.Queue` (lipreal.py:136,161, musereal.py:116,153).  The ring keeps the frame
// bytes in one shared-memory block that the inference process page-locks ONCE, so the device -> host copy of a batch is a single
// asynchronous DMA straight into the slot the consumer will read; only a small descriptor travels through the queue.
extern "C" int mf_host_register(void* p, size_t bytes) {
    MF_REQUIRE(p && bytes > 0, "host_register: null / empty range");
    MF_HIP(hipHostRegister(p, bytes, hipHostRegisterPortable));
    return MF_OK;
}

extern "C" int mf_host_unregister(void* p) {
    MF_REQUIRE(p, "host_unregister: null pointer");
    MF_HIP(hipHostUnregister(p));
    return MF_OK;
}

extern "C" int mf_copy_d2h_async(const void* dev, void* host, size_t bytes, void* stream) {
    MF_REQUIRE(dev && host && bytes > 0, "copy_d2h_async: null / empty argument");
    MF_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return MF_OK;
}

extern "C" int mf_copy_d2h_2d_async(const void* dev, size_t dev_pitch, void* host, size_t host_pitch, size_t width_bytes, size_t rows, void* stream) {
    MF_REQUIRE(dev && host && width_bytes > 0 && rows > 0 && dev_pitch >= width_bytes && host_pitch >= width_bytes, "copy_d2h_2d_async: bad argument");
    MF_HIP(hipMemcpy2DAsync(host, host_pitch, dev, dev_pitch, width_bytes, rows, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return MF_OK;
}

extern "C" int mf_stream_synchronize(void* stream) {
    MF_HIP(hipStreamSynchronize((hipStream_t)stream));
    return MF_OK;
}
