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

#include <atomic>
#include <map>
#include <utility>
// hipFuncAttributeMaxDynamicSharedMemorySize is ONE value per (kernel, device): remember the largest request so far and
// only ever raise it (a later, smaller request must not lower the limit under a launch that still needs the larger one)
int prc_lds_optin(const void* kernel, int bytes) {
    static std::mutex mtx;
    static std::map<std::pair<const void*, int>, int> largest;
    int dev = 0;
    PRC_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mtx);
    const auto key = std::make_pair(kernel, dev);
    auto it = largest.find(key);
    if (it != largest.end() && it->second >= bytes) return PRC_OK;
    PRC_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    largest[key] = bytes;
    return PRC_OK;
}

// ---- roctx ranges (PRC_OPT_MARKERS): the marker library is bound at run time, only when the option is switched on ----
#include <dlfcn.h>
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)(void);
static roctx_push_fn g_roctx_push = nullptr;
static roctx_pop_fn g_roctx_pop = nullptr;
static std::atomic<int> g_markers{0};
static bool roctx_bind() {
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            roctx_push_fn push = (roctx_push_fn)dlsym(h, "roctxRangePushA");
            roctx_pop_fn pop = (roctx_pop_fn)dlsym(h, "roctxRangePop");
            if (push && pop) {
                g_roctx_push = push;
                g_roctx_pop = pop;
                return;
            }
        }
    });
    return g_roctx_push != nullptr;
}
// g_markers is stored with release after roctx_bind() wrote the two function pointers and loaded with acquire here: a
// thread that sees the flag also sees the pointers (the relaxed pair was a formal data race)
PrcRange::PrcRange(const char* name) : on(g_markers.load(std::memory_order_acquire) != 0) {
    if (on) (void)g_roctx_push(name);
}
PrcRange::~PrcRange() {
    if (on) (void)g_roctx_pop();
}

// ---- tuning options (prcore.h: prc_option).  Process-wide values; plans copy what concerns them at creation. ----
static std::atomic<int64_t> g_opt[PRC_OPT_COUNT_];
static const int64_t g_opt_default[PRC_OPT_COUNT_] = {
    /* CAF_MULTI_MODE */ PRC_CAF_MULTI_AUTO, /* CAF_GROUP_MB */ 0, /* LS_TEAM_PIECES */ 32, /* LS_TEAM_ALIGN */ 1,
    /* NLMS_WAVES */ 0, /* LS_CACHE_LIMIT_MB */ 0, /* NLMS_WG_WAVES */ 0, /* CAF_XCD_CONTIG */ 0, /* CAF_PAIR_FRAMES */ 1,
    /* FE_METHOD */ 0, /* CFAR_METHOD */ 0, /* MARKERS */ 0, /* FE_BALANCE */ 0, /* CAF_TEAM8 */ 0, /* FE_FOLD */ 1};
static std::once_flag g_opt_once;
static void opt_init() {
    std::call_once(g_opt_once, [] { for (int i = 0; i < PRC_OPT_COUNT_; ++i) g_opt[i].store(g_opt_default[i]); });
}
int64_t prc_opt(int option) {
    opt_init();
    return g_opt[option].load();
}
extern "C" int prc_set_option(int32_t option, int64_t value) {
    PRC_REQUIRE(option >= 0 && option < PRC_OPT_COUNT_, PRC_EINVAL, "prc_set_option: unknown option %d", option);
    opt_init();
    bool ok = true;
    switch (option) {
        case PRC_OPT_CAF_MULTI_MODE: ok = value >= PRC_CAF_MULTI_AUTO && value <= PRC_CAF_MULTI_PAIRS; break;
        case PRC_OPT_CAF_GROUP_MB: ok = value >= 0; break;
        case PRC_OPT_LS_TEAM_PIECES: ok = value >= 1 && value <= 4096; break;
        case PRC_OPT_LS_TEAM_ALIGN: ok = value == 0 || value == 1; break;
        case PRC_OPT_NLMS_WAVES: ok = value == 0 || value == 1 || value == 2 || value == 4; break;
        case PRC_OPT_LS_CACHE_LIMIT_MB: ok = value >= 0; break;
        case PRC_OPT_NLMS_WG_WAVES: ok = value == 0 || value == 4 || value == 8 || value == 12 || value == 16; break;
        case PRC_OPT_CAF_XCD_CONTIG: ok = value == 0 || value == 1; break;
        case PRC_OPT_CAF_PAIR_FRAMES: ok = value == 0 || value == 1; break;
        case PRC_OPT_FE_METHOD: ok = value >= 0 && value <= 2; break;
        case PRC_OPT_CFAR_METHOD: ok = value == 0 || value == 1; break;
        case PRC_OPT_FE_BALANCE: ok = value >= 0 && value <= 1000; break;
        case PRC_OPT_CAF_TEAM8: ok = value == 0 || value == 1; break;
        case PRC_OPT_FE_FOLD: ok = value == 0 || value == 1; break;
        case PRC_OPT_MARKERS:
            ok = value == 0 || value == 1;
            if (value == 1 && !roctx_bind()) {
                prc_set_error("prc_set_option: PRC_OPT_MARKERS needs librocprofiler-sdk-roctx.so.1 or libroctx64.so.4");
                return PRC_EUNSUPPORTED;
            }
            break;
    }
    PRC_REQUIRE(ok, PRC_EINVAL, "prc_set_option: value %lld out of range for option %d", (long long)value, option);
    g_opt[option].store(value);
    if (option == PRC_OPT_MARKERS) g_markers.store((int)value, std::memory_order_release);
    return PRC_OK;
}
extern "C" int prc_get_option(int32_t option, int64_t* value) {
    PRC_REQUIRE(value && option >= 0 && option < PRC_OPT_COUNT_, PRC_EINVAL, "prc_get_option: bad argument");
    *value = prc_opt(option);
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

extern "C" int prc_mem_info(size_t* free_bytes, size_t* total_bytes) {
    PRC_REQUIRE(free_bytes && total_bytes, PRC_EINVAL, "prc_mem_info: null argument");
    PRC_HIP(hipMemGetInfo(free_bytes, total_bytes));
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
