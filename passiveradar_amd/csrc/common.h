// Shared host/device helpers for libprcore (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <mutex>
#include "../../include/prcore.h"

#define PRC_WAVE 64

void prc_set_error(const char* fmt, ...);

#define PRC_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess) {                                                           \
            prc_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                          __LINE__);                                                       \
            return PRC_EHIP;                                                               \
        }                                                                                  \
    } while (0)

#define PRC_LAUNCH_CHECK()                                                                 \
    do {                                                                                   \
        hipError_t e__ = hipGetLastError();                                                \
        if (e__ != hipSuccess) {                                                           \
            prc_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__),      \
                          __FILE__, __LINE__);                                             \
            return PRC_EHIP;                                                               \
        }                                                                                  \
    } while (0)

#define PRC_REQUIRE(cond, code, ...)                                                       \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            prc_set_error(__VA_ARGS__);                                                    \
            return (code);                                                                 \
        }                                                                                  \
    } while (0)

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Descriptor hand-over (prcore.h, Conventions): the host says how large ITS struct is in the first field; copy what both
// sides know into a zeroed struct of OUR layout, never read past the host's struct, refuse sizes below the version-600
// layout.  `mine` is what the rest of the library reads.
template <class D>
static inline int prc_take_desc(D* mine, const D* theirs, uint32_t min_size, const char* who, const char* what) {
    uint32_t sz, magic;
    memcpy(&sz, theirs, sizeof(sz));                    // the two fields every layout from 600 on has, at offsets 0 and 4
    memcpy(&magic, (const char*)theirs + 4, sizeof(magic));
    if (magic != PRC_DESC_MAGIC) {
        prc_set_error("%s: %s.magic = 0x%08x, not PRC_DESC_MAGIC (0x%08x): the host was built against a prcore.h older than "
                      "version 600 (descriptors now start with struct_size, magic -- PRC_DESC_INIT) and must be rebuilt",
                      who, what, magic, PRC_DESC_MAGIC);
        return PRC_EINVAL;
    }
    if (sz < min_size || (sz & 3u) || sz > (1u << 16)) {
        prc_set_error("%s: %s.struct_size = %u, expected %u (this library, header version %d) or at least %u (version 600): "
                      "set it to sizeof(%s) of the prcore.h the host was built against -- a host built against a pre-600 "
                      "header must be rebuilt", who, what, sz, (unsigned)sizeof(D), PRC_VERSION, min_size, what);
        return PRC_EINVAL;
    }
    memset(mine, 0, sizeof(D));
    memcpy(mine, theirs, sz < sizeof(D) ? sz : sizeof(D));
    mine->struct_size = (uint32_t)sizeof(D);
    return PRC_OK;
}

// hipFuncAttributeMaxDynamicSharedMemorySize for a kernel that wants more than 64 KB of LDS: set once per (kernel, device),
// not on every launch (util.hip)
int prc_lds_optin(const void* kernel, int bytes);
// current value of a prc_option (util.hip); plans read it once, at creation
int64_t prc_opt(int option);
// roctx range around an entry point when PRC_OPT_MARKERS is on (util.hip); otherwise one relaxed atomic load
struct PrcRange {
    bool on;
    explicit PrcRange(const char* name);
    ~PrcRange();
    PrcRange(const PrcRange&) = delete;
    PrcRange& operator=(const PrcRange&) = delete;
};
#define PRC_RANGE(name) PrcRange prc_range__(name)

// ---- device-side complex helpers (float2 = complex64, double2 = complex128) ----
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// acc += p * conj(s)
__device__ __forceinline__ void cmac_conj(float2& acc, float2 p, float2 s) {
    acc.x = fmaf(p.x, s.x, acc.x);
    acc.x = fmaf(p.y, s.y, acc.x);
    acc.y = fmaf(p.y, s.x, acc.y);
    acc.y = fmaf(-p.x, s.y, acc.y);
}
// acc += a * b
__device__ __forceinline__ void cmac(float2& acc, float2 a, float2 b) {
    acc.x = fmaf(a.x, b.x, acc.x);
    acc.x = fmaf(-a.y, b.y, acc.x);
    acc.y = fmaf(a.x, b.y, acc.y);
    acc.y = fmaf(a.y, b.x, acc.y);
}
__device__ __forceinline__ double2 zmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 zconj(double2 a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ double2 zadd(double2 a, double2 b) {
    return make_double2(a.x + b.x, a.y + b.y);
}
__device__ __forceinline__ double2 zsub(double2 a, double2 b) {
    return make_double2(a.x - b.x, a.y - b.y);
}
__device__ __forceinline__ double2 zscale(double2 a, double s) {
    return make_double2(a.x * s, a.y * s);
}
// a / b (b != 0), plain formula in double
__device__ __forceinline__ double2 zdiv(double2 a, double2 b) {
    double d = b.x * b.x + b.y * b.y;
    return make_double2((a.x * b.x + a.y * b.y) / d, (a.y * b.x - a.x * b.y) / d);
}

// ---- raw buffer loads: uniform 64-bit base + num_records in SGPRs, one 32-bit lane offset in a
// VGPR, and the hardware range check returns 0 for every lane beyond num_records -- so zero
// padding, ragged tails and "past the end" prefetches cost no compare/select/address VALU at all.
typedef unsigned int prc_v2u __attribute__((ext_vector_type(2)));
// base and bytes MUST be wave-uniform; the readfirstlane pins them in SGPRs (hipcc otherwise keeps
// min/max chains in VGPRs and wraps every load in a waterfall loop).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t prc_rsrc(const void* base, unsigned bytes) {
    const unsigned long long p = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
    const unsigned nb = __builtin_amdgcn_readfirstlane(bytes);
    void* q = (void*)(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, (short)0, (int)nb, 0x00020000);
}
__device__ __forceinline__ float2 prc_buf_load_c64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const prc_v2u x = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
    return make_float2(__uint_as_float(x.x), __uint_as_float(x.y));
}
// two adjacent complex64 per lane in one 16-byte access (1 KB per wavefront instruction)
typedef unsigned int prc_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void prc_buf_load_2c64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float2& a, float2& b) {
    const prc_v4u x = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    a = make_float2(__uint_as_float(x.x), __uint_as_float(x.y));
    b = make_float2(__uint_as_float(x.z), __uint_as_float(x.w));
}
__device__ __forceinline__ void prc_buf_store_c64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float2 v) {
    prc_v2u x;
    x.x = __float_as_uint(v.x);
    x.y = __float_as_uint(v.y);
    __builtin_amdgcn_raw_buffer_store_b64(x, r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ float prc_buf_load_f32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}

// The reference's frequency_shift keeps the sample index in complex64, so the phase ramp is
// float32:  ph = fl32(fl32(a32 * n) * rcp32)  (signal_utils.py:24-27; NumPy complex/real
// division multiplies by the float32 reciprocal).  No FMA contraction is possible here
// (two multiplies), and the file is built without fast-math.
struct PhaseRamp {
    float a32;    // fl32(2*pi*fc)
    float rcp32;  // fl32(1 / fl32(Fs))
    float off32;  // fl32(phase_offset)
    int enabled;  // 0: no rotation (fc == 0 and offset == 0)
};
__device__ __forceinline__ float2 phase_rot(const PhaseRamp& pr, int64_t n) {
    float ph = (pr.a32 * (float)n) * pr.rcp32;
    ph = ph + pr.off32;
    float s, c;
    sincosf(ph, &s, &c);
    return make_float2(c, s);
}
