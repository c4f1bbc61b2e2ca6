// Front end on device (SURVEY 8f "next" #1): the per-block chain of the reference's main.py:105-166
//   deinterleave_IQ (signal_utils.py:19-22)  ->  frequency_shift with a block phase (:24-27, main.py:125-149)
//   ->  resample = scipy.signal.resample_poly(x, up, dn, padtype='line') (:15-17)
// fused into ONE kernel: raw interleaved scalars in, IF complex64 samples out, nothing in between
// touches HBM (the reference materialises a complex64 and two complex128 arrays per block).
//
// Polyphase form of resample_poly (the same closed form the CPU checker restates and pins):
//   y[m] = sum_j h[(t mod up) + up j] xe[t div up - j],   t = (m + n_pre_remove) dn,
// h = firwin(20 max(up,dn)+1, 1/max(up,dn), ('kaiser',5.0)) * up, zero-padded in front (host side,
// scipy), xe = the tuned block extended linearly through its first and last sample (upfirdn 'line').
// Tuning keeps the reference's arithmetic: the phase ramp is float32 (sample index held as
// complex64), the block phase is added and the exponential taken in double (array phase_offset
// promotes to complex128) -- at 100 kHz offset the float32 ramp is quantised to 1/16 rad, so this
// has to be reproduced, not improved.
#include "common.h"
#include <math.h>
#include <vector>

#define FE_THREADS 256


struct FeArgs {
    const void* raw;
    float2* out;
    const float* taps;        // [J][up] polyphase layout: taps[j*up + p] = h[p + up*j]
    const double* phases;     // per block phase offset (device), or nullptr
    int64_t raw_stride;       // elements of the raw type between blocks (complex elements for C64)
    int64_t out_stride;
    int64_t n_in;             // complex samples per block
    int64_t n_out;
    int32_t up, dn, J, n_pre_remove;
    int32_t mix;              // apply the frequency shift
    int32_t opw;              // outputs per workgroup (<= FE_THREADS; fewer when the decimation ratio is large)
    PhaseRamp pr;
};

template <int SRC>
__device__ __forceinline__ float2 fe_load(const void* raw, int64_t i) {
    if (SRC == PRC_RAW_I8) {
        const signed char* p = (const signed char*)raw;
        return make_float2((float)p[2 * i], (float)p[2 * i + 1]);
    } else if (SRC == PRC_RAW_U8) {
        const unsigned char* p = (const unsigned char*)raw;
        return make_float2((float)p[2 * i], (float)p[2 * i + 1]);
    } else if (SRC == PRC_RAW_I16) {
        const short* p = (const short*)raw;
        return make_float2((float)p[2 * i], (float)p[2 * i + 1]);
    } else if (SRC == PRC_RAW_F32) {
        const float* p = (const float*)raw;
        return make_float2(p[2 * i], p[2 * i + 1]);
    } else {
        return ((const float2*)raw)[i];
    }
}

// tuned sample i of the block, in double (reference: complex128 after the array phase offset)
template <int SRC>
__device__ __forceinline__ double2 fe_tuned(const FeArgs& a, const void* raw, int64_t i, double blk_phase) {
    const float2 v = fe_load<SRC>(raw, i);
    if (!a.mix) return make_double2(v.x, v.y);
    const float ph32 = (a.pr.a32 * (float)i) * a.pr.rcp32;      // float32 ramp, as the reference
    double s, c;
    sincos((double)ph32 + blk_phase, &s, &c);
    return make_double2((double)v.x * c - (double)v.y * s, (double)v.x * s + (double)v.y * c);
}

template <int SRC>
__global__ __launch_bounds__(FE_THREADS) void frontend_kernel(FeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* H = reinterpret_cast<float*>(smem_raw);                  // J * up
    float2* X = reinterpret_cast<float2*>(H + ((a.J * a.up + 1) & ~1));   // staged tuned inputs
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const void* raw;
    if (SRC == PRC_RAW_I8 || SRC == PRC_RAW_U8) raw = (const char*)a.raw + (int64_t)b * a.raw_stride;
    else if (SRC == PRC_RAW_I16) raw = (const short*)a.raw + (int64_t)b * a.raw_stride;
    else if (SRC == PRC_RAW_F32) raw = (const float*)a.raw + (int64_t)b * a.raw_stride;
    else raw = (const float2*)a.raw + (int64_t)b * a.raw_stride;
    const double blk_phase = (a.mix && a.phases) ? a.phases[b] : 0.0;
    for (int i = tid; i < a.J * a.up; i += FE_THREADS) H[i] = a.taps[i];

    // outputs of this workgroup and the input span they touch
    const int64_t m0 = (int64_t)blockIdx.x * a.opw;
    int64_t m1 = m0 + a.opw;
    if (m1 > a.n_out) m1 = a.n_out;
    const int64_t i_hi = ((m1 - 1 + a.n_pre_remove) * a.dn) / a.up;      // newest input of the last output
    const int64_t i_lo = ((m0 + a.n_pre_remove) * a.dn) / a.up - (a.J - 1);
    const int span = (int)(i_hi - i_lo + 1);
    // 'line' extension: xe[i] = x[0] + i*slope (i < 0),  x[n-1] + (i-(n-1))*slope (i >= n)
    const double2 x0 = fe_tuned<SRC>(a, raw, 0, blk_phase);
    const double2 xl = fe_tuned<SRC>(a, raw, a.n_in - 1, blk_phase);
    const double inv = a.n_in > 1 ? 1.0 / (double)(a.n_in - 1) : 0.0;
    const double2 slope = make_double2((xl.x - x0.x) * inv, (xl.y - x0.y) * inv);
    for (int k = tid; k < span; k += FE_THREADS) {
        const int64_t i = i_lo + k;
        double2 v;
        if (i < 0) v = make_double2(x0.x + (double)i * slope.x, x0.y + (double)i * slope.y);
        else if (i >= a.n_in) {
            const double d = (double)(i - (a.n_in - 1));
            v = make_double2(xl.x + d * slope.x, xl.y + d * slope.y);
        } else v = fe_tuned<SRC>(a, raw, i, blk_phase);
        X[k] = make_float2((float)v.x, (float)v.y);
    }
    __syncthreads();
    const int64_t m = m0 + tid;
    if (tid >= a.opw || m >= a.n_out) return;
    const int64_t t = (m + a.n_pre_remove) * a.dn;
    const int phase = (int)(t % a.up);
    const int base = (int)(t / a.up - i_lo);          // X index of the newest input of this output
    float2 acc = make_float2(0.f, 0.f);
    const float* Hp = H + phase;
#pragma unroll 4
    for (int j = 0; j < a.J; ++j) {
        const float hj = Hp[j * a.up];
        const float2 x = X[base - j];
        acc.x = fmaf(hj, x.x, acc.x);
        acc.y = fmaf(hj, x.y, acc.y);
    }
    a.out[(int64_t)b * a.out_stride + m] = acc;
}

struct prc_frontend_plan {
    prc_frontend_desc desc;
    float* d_taps = nullptr;     // polyphase layout
    double* d_phases = nullptr;  // max_blocks
    int J = 0;
    int64_t n_in = 0, n_out = 0;
    std::mutex mtx;
};

extern "C" int prc_frontend_plan_destroy(prc_frontend_plan* p) {
    if (!p) return PRC_OK;
    if (p->d_taps) (void)hipFree(p->d_taps);
    if (p->d_phases) (void)hipFree(p->d_phases);
    delete p;
    return PRC_OK;
}

extern "C" int prc_frontend_plan_create(prc_frontend_plan** plan, const prc_frontend_desc* d) {
    PRC_REQUIRE(plan && d, PRC_EINVAL, "prc_frontend_plan_create: null argument");
    PRC_REQUIRE(d->n_in > 1 && d->up > 0 && d->down > 0 && d->ntaps > 0 && d->taps_host && d->max_blocks > 0,
                PRC_EINVAL, "prc_frontend_plan_create: bad size");
    PRC_REQUIRE(d->raw_dtype >= PRC_RAW_I8 && d->raw_dtype <= PRC_RAW_C64, PRC_EINVAL,
                "prc_frontend_plan_create: unknown raw dtype %d", d->raw_dtype);
    prc_frontend_plan* p = new prc_frontend_plan();
    p->desc = *d;
    p->desc.taps_host = nullptr;
    p->n_in = d->n_in;
    int64_t no = d->n_in * d->up;
    p->n_out = no / d->down + (no % d->down ? 1 : 0);
    p->J = (d->ntaps + d->up - 1) / d->up;
    std::vector<float> poly((size_t)p->J * d->up, 0.f);
    for (int k = 0; k < d->ntaps; ++k) poly[(size_t)(k / d->up) * d->up + (k % d->up)] = d->taps_host[k];
    hipError_t e = hipMalloc(&p->d_taps, sizeof(float) * poly.size());
    if (e == hipSuccess) e = hipMemcpy(p->d_taps, poly.data(), sizeof(float) * poly.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&p->d_phases, sizeof(double) * d->max_blocks);
    if (e != hipSuccess) {
        prc_set_error("prc_frontend_plan_create: device setup failed: %s", hipGetErrorString(e));
        prc_frontend_plan_destroy(p);
        return PRC_EHIP;
    }
    *plan = p;
    return PRC_OK;
}

extern "C" int prc_frontend_out_len(const prc_frontend_plan* p, int64_t* n_out) {
    PRC_REQUIRE(p && n_out, PRC_EINVAL, "prc_frontend_out_len: null argument");
    *n_out = p->n_out;
    return PRC_OK;
}

extern "C" int prc_frontend_execute(prc_frontend_plan* p, const void* raw, int64_t raw_stride, int32_t mix,
                                    double fc, double fs, const double* phases_host, void* out,
                                    int64_t out_stride, int32_t nblocks, void* stream_) {
    PRC_REQUIRE(p && raw && out, PRC_EINVAL, "prc_frontend_execute: null argument");
    PRC_REQUIRE(nblocks > 0 && nblocks <= p->desc.max_blocks, PRC_EINVAL,
                "prc_frontend_execute: nblocks=%d outside [1, %d]", nblocks, p->desc.max_blocks);
    PRC_REQUIRE(out_stride >= p->n_out, PRC_ESHAPE, "prc_frontend_execute: out_stride shorter than the output");
    hipStream_t stream = (hipStream_t)stream_;
    std::lock_guard<std::mutex> lk(p->mtx);
    FeArgs a;
    a.raw = raw;
    a.out = (float2*)out;
    a.taps = p->d_taps;
    a.phases = nullptr;
    if (mix && phases_host) {
        PRC_HIP(hipMemcpyAsync(p->d_phases, phases_host, sizeof(double) * nblocks, hipMemcpyHostToDevice, stream));
        a.phases = p->d_phases;
    }
    a.raw_stride = raw_stride;
    a.out_stride = out_stride;
    a.n_in = p->n_in;
    a.n_out = p->n_out;
    a.up = p->desc.up;
    a.dn = p->desc.down;
    a.J = p->J;
    a.n_pre_remove = p->desc.n_pre_remove;
    a.mix = mix ? 1 : 0;
    a.pr.a32 = (float)(2.0 * 3.14159265358979323846 * fc);
    a.pr.rcp32 = 1.0f / (float)fs;
    a.pr.off32 = 0.f;
    a.pr.enabled = a.mix;
    // staged span per workgroup: opw outputs * dn/up inputs + J taps of history; opw shrinks (in steps of one
    // wavefront) until the span fits the CU's LDS, so any decimation ratio runs
    const size_t lds_taps = sizeof(float) * ((size_t)(a.J * a.up + 1) & ~(size_t)1);
    a.opw = FE_THREADS;
    size_t lds = 0;
    for (;;) {
        const int64_t span = ((int64_t)a.opw * a.dn) / a.up + a.J + 4;
        lds = lds_taps + sizeof(float2) * (size_t)span;
        if (lds <= 150 * 1024 || a.opw <= 1) break;
        a.opw = a.opw > 64 ? a.opw - 64 : a.opw / 2;
    }
    PRC_REQUIRE(lds <= 160 * 1024, PRC_EUNSUPPORTED, "prc_frontend_execute: resampling ratio %d/%d needs %zu B of LDS",
                a.up, a.dn, lds);
    dim3 grid((unsigned)ceil_div64(p->n_out, a.opw), (unsigned)nblocks);
#define PRC_FE_CASE(S)                                                                               \
    case S:                                                                                          \
        (void)hipFuncSetAttribute((const void*)frontend_kernel<S>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL(frontend_kernel<S>, grid, dim3(FE_THREADS), lds, stream, a);              \
        break;
    switch (p->desc.raw_dtype) {
        PRC_FE_CASE(PRC_RAW_I8)
        PRC_FE_CASE(PRC_RAW_U8)
        PRC_FE_CASE(PRC_RAW_I16)
        PRC_FE_CASE(PRC_RAW_F32)
        PRC_FE_CASE(PRC_RAW_C64)
    }
#undef PRC_FE_CASE
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}

// deinterleave_IQ alone (signal_utils.py:19-22): raw scalars -> complex64
template <int SRC>
__global__ void deinterleave_kernel(const void* raw, float2* out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = fe_load<SRC>(raw, i);
}

extern "C" int prc_deinterleave(const void* raw, int32_t raw_dtype, int64_t n_complex, void* out, void* stream) {
    PRC_REQUIRE(raw && out && n_complex > 0, PRC_EINVAL, "prc_deinterleave: bad argument");
    int64_t blocks = ceil_div64(n_complex, 256);
    if (blocks > 4096) blocks = 4096;
    switch (raw_dtype) {
        case PRC_RAW_I8: hipLaunchKernelGGL(deinterleave_kernel<PRC_RAW_I8>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, raw, (float2*)out, n_complex); break;
        case PRC_RAW_U8: hipLaunchKernelGGL(deinterleave_kernel<PRC_RAW_U8>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, raw, (float2*)out, n_complex); break;
        case PRC_RAW_I16: hipLaunchKernelGGL(deinterleave_kernel<PRC_RAW_I16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, raw, (float2*)out, n_complex); break;
        case PRC_RAW_F32: hipLaunchKernelGGL(deinterleave_kernel<PRC_RAW_F32>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, raw, (float2*)out, n_complex); break;
        default: prc_set_error("prc_deinterleave: unknown raw dtype %d", raw_dtype); return PRC_EINVAL;
    }
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}

// frequency_shift with an ARRAY phase offset (main.py:133-149): float32 ramp + double block phase,
// exponential in double, complex128 out (the reference's promotion).
__global__ void freq_shift_block_kernel(const float2* __restrict__ x, double2* __restrict__ y, int64_t n,
                                        PhaseRamp pr, double blk_phase) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float ph32 = (pr.a32 * (float)i) * pr.rcp32;
        double s, c;
        sincos((double)ph32 + blk_phase, &s, &c);
        const float2 v = x[i];
        y[i] = make_double2((double)v.x * c - (double)v.y * s, (double)v.x * s + (double)v.y * c);
    }
}

extern "C" int prc_frequency_shift_block(const void* x, void* y, int64_t n, double fc, double fs,
                                         double block_phase, void* stream) {
    PRC_REQUIRE(x && y && n > 0 && fs != 0.0, PRC_EINVAL, "prc_frequency_shift_block: bad argument");
    PhaseRamp pr;
    pr.a32 = (float)(2.0 * 3.14159265358979323846 * fc);
    pr.rcp32 = 1.0f / (float)fs;
    pr.off32 = 0.f;
    pr.enabled = 1;
    int64_t blocks = ceil_div64(n, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(freq_shift_block_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)x, (double2*)y, n, pr, block_phase);
    PRC_LAUNCH_CHECK();
    return PRC_OK;
}
