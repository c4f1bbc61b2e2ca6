// Library-level entry points: version, thread-local error text, device + memory helpers.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void prc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#include <set>
#include <utility>
int prc_lds_optin(const void* kernel, int bytes) {
    static std::mutex mtx;
    static std::set<std::pair<std::pair<const void*, int>, int>> sizes;      // (kernel, device, bytes) already set
    int dev = 0;
    PRC_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mtx);
    const auto key = std::make_pair(std::make_pair(kernel, dev), bytes);
    if (sizes.count(key)) return PRC_OK;
    PRC_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    sizes.insert(key);
    return PRC_OK;
}

extern "C" int prc_version(void) { return PRC_VERSION; }

extern "C" const char* prc_last_error(void) { return g_err; }

extern "C" int prc_device_count(int* count) {
    PRC_REQUIRE(count, PRC_EINVAL, "prc_device_count: null argument");
    *count = 0;
    PRC_HIP(hipGetDeviceCount(count));
    return PRC_OK;
}

extern "C" int prc_set_device(int device) {
    PRC_HIP(hipSetDevice(device));
    return PRC_OK;
}

extern "C" int prc_get_device(int* device) {
    PRC_REQUIRE(device, PRC_EINVAL, "prc_get_device: null argument");
    PRC_HIP(hipGetDevice(device));
    return PRC_OK;
}

extern "C" int prc_malloc(void** dptr, size_t bytes) {
    PRC_REQUIRE(dptr, PRC_EINVAL, "prc_malloc: null argument");
    *dptr = nullptr;
    if (bytes == 0) return PRC_OK;
    PRC_HIP(hipMalloc(dptr, bytes));
    return PRC_OK;
}

extern "C" int prc_free(void* dptr) {
    if (dptr) PRC_HIP(hipFree(dptr));
    return PRC_OK;
}

extern "C" int prc_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
    if (bytes == 0) return PRC_OK;
    PRC_REQUIRE(dst && src, PRC_EINVAL, "prc_memcpy_h2d: null argument");
    PRC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return PRC_OK;
}

extern "C" int prc_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
    if (bytes == 0) return PRC_OK;
    PRC_REQUIRE(dst && src, PRC_EINVAL, "prc_memcpy_d2h: null argument");
    PRC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return PRC_OK;
}

extern "C" int prc_memset(void* dptr, int value, size_t bytes, void* stream) {
    if (bytes == 0) return PRC_OK;
    PRC_REQUIRE(dptr, PRC_EINVAL, "prc_memset: null argument");
    PRC_HIP(hipMemsetAsync(dptr, value, bytes, (hipStream_t)stream));
    return PRC_OK;
}

extern "C" int prc_stream_sync(void* stream) {
    PRC_HIP(hipStreamSynchronize((hipStream_t)stream));
    return PRC_OK;
}
